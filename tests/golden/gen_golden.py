#!/usr/bin/env python3
"""Generates tests/golden/*.json from the reference's own sqllogictests and bundled data.

Run in the dev container only (needs /root/reference; the GPU box does not have it):
    python tests/golden/gen_golden.py

Every fixture records the reference file and line range its expectation block was cut from.
Expectation rows are extracted mechanically (the text after a `----` line up to the next blank line);
graph inputs are either the INSERT tuples of the same test file (parsed from the cited line) or the
bundled LDBC SNB SF0.003 parquet files (vertex rowid = row position in the vertex file, edge rowid = row
position in the edge file — SURVEY.md Appendix D).
"""
import json
import os
import re

import pyarrow.parquet as pq

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def block(path, sep_line):
    """Rows following the `----` at 1-based line `sep_line`, split on tabs."""
    lines = open(os.path.join(REF, path)).read().split("\n")
    assert lines[sep_line - 1].strip() == "----", (path, sep_line, lines[sep_line - 1])
    rows = []
    i = sep_line
    while i < len(lines) and lines[i].strip() != "":
        rows.append(lines[i].split("\t"))
        i += 1
    return rows, "%s:%d-%d" % (path, sep_line + 1, i)


def tuples_at(path, line_no, table):
    """(a,b,c) integer tuples of `INSERT INTO <table> VALUES ...` on the given 1-based line."""
    line = open(os.path.join(REF, path)).read().split("\n")[line_no - 1]
    m = re.search(r"INSERT INTO %s VALUES (.*?);" % table, line, flags=re.I)
    assert m, (path, line_no)
    return [[int(x) for x in t.split(",")] for t in re.findall(r"\(([^)]*)\)", m.group(1))]


def plist(text):
    text = text.strip()
    assert text[0] == "[" and text[-1] == "]"
    inner = text[1:-1].strip()
    return [int(x) for x in inner.split(",")] if inner else []


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
    print("wrote", name)


NAMES = {"Daniel": 0, "Tavneet": 1, "Gabor": 2, "Peter": 3, "David": 4}  # Student rowids (INSERT order)


def student_directed():
    t = "test/sql/path_finding/shortest_path.test"
    know = tuples_at(t, 14, "know")  # 8 edges (src, dst, createDate); Student ids == rowids 0..4
    rows, cite = block(t, 66)  # :59-82, ANY SHORTEST {1,3}, all pairs
    exp = [{"src": NAMES[r[2]], "dst": NAMES[r[3]], "length": int(r[0]), "path": plist(r[1])} for r in rows]
    rows2, cite2 = block(t, 125)  # :96-128 hand-written UDF SQL, Daniel only
    exp2 = [{"src": NAMES[r[1]], "dst": NAMES[r[2]], "path": plist(r[0])} for r in rows2]
    dump("student_directed.json", {
        "source": cite, "V": 5, "edges": [[e[0], e[1]] for e in know],
        "lower": 1, "upper": 3, "paths": exp,
        "udf_sql": {"source": cite2, "src_filter": 0, "paths": exp2},
    })


def student_csr_layout():
    t = "test/sql/scalar/getpgschema.test"
    know = tuples_at(t, 20, "know")  # 9 edges
    e_rows, cite_e = block(t, 87)
    v_rows, cite_v = block(t, 100)
    dump("student_csr_layout.json", {
        "source": cite_e + " ; " + cite_v, "V": 5, "edges": [[e[0], e[1]] for e in know],
        "csr_e": [int(r[0]) for r in e_rows], "csr_v": [int(r[0]) for r in v_rows],
    })


def student_undirected():
    t = "test/sql/path_finding/undirected_paths.test"
    know = tuples_at(t, 11, "know")  # 9 edges
    allp, cite = block(t, 98)  # :91-123 all pairs, unbounded
    from0, cite0 = block(t, 30)
    from4, cite4 = block(t, 44)
    bounded, citeb = block(t, 151)  # {0,2} from 0
    dump("student_undirected.json", {
        "source": cite, "V": 5, "edges": [[e[0], e[1]] for e in know],
        "all_pairs": [[int(x) for x in r] for r in allp],
        "from0": {"source": cite0, "rows": [[int(x) for x in r] for r in from0]},
        "from4": {"source": cite4, "rows": [[int(x) for x in r] for r in from4]},
        "bounded_0_2_from0": {"source": citeb, "rows": [[int(x) for x in r] for r in bounded]},
    })


def edgeless():
    t = "test/sql/path_finding/edgeless_graph.test"
    rows, cite = block(t, 31)
    dump("edgeless.json", {
        "source": cite, "V": 3, "edges": [],
        "rows": [{"src_id": int(r[0]), "dst_id": int(r[1]), "path": plist(r[2]), "length": int(r[4])} for r in rows],
        "note": "node ids 1,2,3 have rowids 0,1,2; only zero-length paths exist",
    })


def pairs_vertices_only():
    t = "test/sql/create_pg/all_properties.test"
    rows, cite = block(t, 70)
    # graph of that test: Student 0..3?, read the INSERT for know
    txt = open(os.path.join(REF, t)).read().split("\n")
    know_line = next(i for i, l in enumerate(txt, 1) if re.search(r"INSERT INTO know", l, flags=re.I))
    know = tuples_at(t, know_line, "know")
    stud_line = next(i for i, l in enumerate(txt, 1) if re.search(r"INSERT INTO Student", l, flags=re.I))
    m = re.search(r"INSERT INTO Student VALUES (.*?);", txt[stud_line - 1], flags=re.I)
    ids = [int(t_.split(",")[0]) for t_ in re.findall(r"\(([^)]*)\)", m.group(1))]
    dump("all_properties_vertices.json", {
        "source": cite, "student_ids": ids, "edges": [[e[0], e[1]] for e in know],
        "graph_source": "%s:%d,%d" % (t, stud_line, know_line),
        "rows": [{"src_id": int(r[0]), "dst_id": int(r[1]), "vertices": plist(r[2])} for r in rows],
    })


def snb():
    person = pq.read_table(os.path.join(REF, "data/SNB0.003/person.parquet")).column("id").to_pylist()
    knows = pq.read_table(os.path.join(REF, "data/SNB0.003/person_knows_person.parquet"))
    p1 = knows.column("Person1Id").to_pylist()
    p2 = knows.column("Person2Id").to_pylist()
    rowid = {pid: i for i, pid in enumerate(person)}
    edges = [[rowid[a], rowid[b]] for a, b in zip(p1, p2)]  # edge rowid = position
    t = "test/sql/path_finding/complex_matching.test"
    rows, cite = block(t, 334)  # :329-360, from person id 16 (rowid 16), {1,3}
    from16 = [{"src": rowid[int(r[1])], "dst": rowid[int(r[2])], "path": plist(r[0])} for r in rows]
    rows4, cite4 = block(t, 154)  # :114-200 hand-written UDF SQL from id 28587302322180, between 1 and 3
    from4 = sorted({(rowid[28587302322180], int(r[2]), tuple(plist(r[0]))) for r in rows4})
    ic13, cite13 = block("test/sql/snb/snb.test", 113)
    ic13u, cite13u = block("test/sql/snb/snb_inheritance.test", 92)
    dump("snb003_knows.json", {
        "source": cite, "V": len(person), "person_ids": person, "edges": edges,
        "data_source": "data/SNB0.003/person.parquet, person_knows_person.parquet",
        "from16_1_3": from16,
        "from4_1_3": {"source": cite4, "paths": [{"src": s, "dst": d, "path": list(p)} for s, d, p in from4]},
        "ic13_directed": {"source": cite13, "src": rowid[int(ic13[0][1])], "dst": rowid[int(ic13[0][2])],
                          "length": int(ic13[0][0])},
        "ic13_undirected": {"source": cite13u, "src": rowid[int(ic13u[0][1])], "dst": rowid[int(ic13u[0][2])],
                            "length": int(ic13u[0][0])},
    })
    # reply graph (used as a small weighted-path input; no reference expectation exists for it)
    msg = pq.read_table(os.path.join(REF, "data/SNB0.003/message.parquet")).column("id").to_pylist()
    rep = pq.read_table(os.path.join(REF, "data/SNB0.003/message_replyof_message.parquet"))
    cols = rep.column_names
    mrow = {m: i for i, m in enumerate(msg)}
    a = rep.column(cols[-2]).to_pylist()
    b = rep.column(cols[-1]).to_pylist()
    dump("snb003_replyof.json", {
        "source": "data/SNB0.003/message.parquet, message_replyof_message.parquet (columns %s)" % cols,
        "V": len(msg), "edges": [[mrow[x], mrow[y]] for x, y in zip(a, b)],
        "note": "input only: cheapest_path_length has no expectation in the reference tests (parity unpinned)",
    })


def csr_segfault():
    t = "test/sql/csr_segfault.test"
    v, cv = block(t, 50)
    e, ce = block(t, 55)
    dump("csr_segfault.json", {"source": cv + " ; " + ce, "V": 5000, "count_v": int(v[0][0]), "count_e": int(e[0][0]),
                               "note": "know = student positional join student: edge i -> i for i in 0..4999"})


def w_type():
    t = "test/sql/scalar/get_csr_w_type.test"
    txt = open(os.path.join(REF, t)).read().split("\n")
    seps = [i for i, l in enumerate(txt, 1) if l.strip() == "----"]
    vals = []
    for s in seps:
        if txt[s - 2].strip().lower().startswith("select csr_get_w_type") and s < len(txt) and txt[s].strip().isdigit():
            vals.append({"query": txt[s - 2].strip(), "value": int(txt[s].strip()), "line": s + 1})
    dump("csr_w_type.json", {"source": t, "cases": vals,
                             "note": "weight literal 12 -> type 1 (int64), 1.2 -> type 2 (double), none -> 0"})


def analytics():
    """local_clustering_coefficient / pagerank / weakly_connected_component expectations."""
    t = "test/sql/scalar/local_clustering_coefficient.test"
    know = tuples_at(t, 57, "know")
    rows, cite = block(t, 75)
    person = pq.read_table(os.path.join(REF, "data/SNB0.003/person.parquet")).column("id").to_pylist()
    rowid = {pid: i for i, pid in enumerate(person)}
    srows, scite = block(t, 131)
    dump("lcc.json", {
        "student": {"source": cite, "V": 5, "edges": [[e[0], e[1]] for e in know],
                    "rows": [[int(r[0]), r[1]] for r in rows]},
        "snb003": {"source": scite, "graph": "snb003_knows.json (undirected feed)",
                   "rows": [[rowid[int(r[0])], r[1]] for r in srows]},
        "note": "FLOAT column: DuckDB prints the shortest decimal that round-trips a float32",
    })
    t = "test/sql/scalar/pagerank.test"
    know = tuples_at(t, 11, "know")
    rows, cite = block(t, 25)
    txt = open(os.path.join(REF, t)).read().split("\n")
    know2 = [[int(x) for x in re.findall(r"-?\d+", txt[i])][:2] for i in range(54, 68)]  # lines 55-68: (src, dst, edge)
    rows2, cite2 = block(t, 82)
    dump("pagerank.json", {
        "g1": {"source": cite, "V": 5, "edges": [[e[0], e[1]] for e in know], "rows": [[int(r[0]), r[1]] for r in rows]},
        "g2": {"source": cite2, "V": 5, "edges": know2, "rows": [[int(r[0]), r[1]] for r in rows2],
               "graph_source": "%s:55-68" % t},
        "note": "directed CSR feed; DOUBLE column printed with 17 significant digits",
    })
    t = "test/sql/scalar/weakly_connected_component.test"
    cases = []
    for vline, eline, sep, V in ((10, 13, 31, 5), (40, 44, 59, 5), (68, 72, 86, 6), (97, 101, 115, 5), (124, 128, 142, 5)):
        rows, cite = block(t, sep)
        cases.append({"source": cite, "V": V, "edges": [[e[0], e[1]] for e in tuples_at(t, eline, "know")],
                      "rows": [[int(r[0]), int(r[1])] for r in rows]})
    dump("wcc.json", {"cases": cases, "note": "undirected feed (CreateUndirectedCSRCTE); the component id is the root the "
                                              "reference's sequential union-find ends in, not a canonical label"})


if __name__ == "__main__":
    analytics()
    student_directed()
    student_csr_layout()
    student_undirected()
    edgeless()
    pairs_vertices_only()
    snb()
    csr_segfault()
    w_type()
