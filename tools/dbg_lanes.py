import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import duckpgq_extension_amd as pgq
from oracle.pgq_oracle import OracleCSR

def random_graph(rng, V, E, skew=False):
    if skew:
        s = (rng.random(E) ** 3 * V).astype(np.int64)
        d = (rng.random(E) ** 2 * V).astype(np.int64)
    else:
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    return s, d, np.arange(E, dtype=np.int64)

for words in (1, 2, 4, 8, 32):
    rng = np.random.default_rng(100 * words)
    V, E = 3000, 14000
    s, d, e = random_graph(rng, V, E, skew=True)
    st = pgq.PgqState(); st.build_csr(0, V, s, d, e, None)
    ora = OracleCSR.from_edges(V, s, d, e, None)
    n = 1500
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    oln, ook = ora.lean_iterativelength(V, ps, pd)
    pgq.set_option("words", words); pgq.set_option("hub_chunk", 64); pgq.set_option("push_chunk", 64)
    for lanes, fp, lds, un, probe in ((0,1,1,2,0),(1,1,1,2,0),(1,1,0,2,0),(1,1,1,4,0),(1,1,1,1,0),(1,0,1,4,1)):
        pgq.set_option("lanes", lanes); pgq.set_option("force_pull", fp); pgq.set_option("sparse_lds", lds)
        pgq.set_option("lanes_unroll", un); pgq.set_option("probe", probe); pgq.set_option("streams", 1)
        ln, ok = st.iterativelength(0, V, ps, pd)
        bad = int(((ok != ook) | ((ln != oln) & ok & ook)).sum())
        print("words", words, "lanes", lanes, "force_pull", fp, "lds", lds, "un", un, "probe", probe, "mismatches", bad)
