import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import duckpgq_extension_amd as pgq
from oracle.pgq_oracle import OracleCSR
seed = 500016
def run(over):
    rng = np.random.default_rng(seed)
    V = int(rng.choice([300, 3000, 20000])); E = int(V * float(rng.choice([1.5, 5, 14])))
    if rng.random() < 0.5:
        s = (rng.random(E) ** 3 * V).astype(np.int64); d = (rng.random(E) ** 2 * V).astype(np.int64)
    else:
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    e = np.arange(E, dtype=np.int64)
    wkind = int(rng.integers(0, 3))
    w = [rng.integers(1, 1000, E), rng.integers(0, 4, E), rng.random(E) + 0.01][wkind]
    opts = {"meet": int(rng.integers(0, 2)), "meet_bias": 1e9, "paths_reserve_mb": int(rng.choice([0, 1024])), "meet_cap_paths": int(rng.choice([300, 1 << 14])),
            "meet4_lds_kb": int(rng.choice([0, 150])), "relax_bidir": int(rng.integers(0, 2)), "relax_light": int(rng.choice([0, 2])),
            "relax_labels32": int(rng.integers(0, 2)), "chain": int(rng.integers(0, 2)), "streams": int(rng.choice([1, 3])),
            "relax_bidir_c0_div": int(rng.choice([1, 64, 1 << 20])), "meet_spin_wait": int(rng.integers(0, 2))}
    opts.update(over)
    for k, v in opts.items(): pgq.set_option(k, v)
    st = pgq.PgqState(); st.build_csr(0, V, s, d, e, w); ora = OracleCSR.from_edges(V, s, d, e, w)
    n = int(rng.choice([1, 64, 700, 2500]))
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    if rng.random() < 0.3: ps = ps[rng.integers(0, max(1, n // 50), n)]
    bad = 0
    want = ora.lean_shortestpath(V, ps, pd)
    for k in range(5):
        got = st.shortestpath(0, V, ps, pd)
        diff = [i for i in range(n) if got[i] != want[i]]
        bad += len(diff)
        if diff and k == 0: print("  first diffs", diff[:3], [got[i] for i in diff[:2]], [want[i] for i in diff[:2]])
    print(over, "n", n, "mismatching rows over 5 calls:", bad)
for over in ({}, {"paths_reserve_mb": 1024}, {"meet_spin_wait": 0}, {"paths_reserve_mb": 1024, "meet_spin_wait": 0}, {"meet4_lds_kb": 150}):
    run(over)
