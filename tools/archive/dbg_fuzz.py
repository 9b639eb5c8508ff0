import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import fuzz_parity
import extractorb_amd as X
from test_gpu_parity import oracle_run
rng = np.random.default_rng(103)
for t in range(8):
    c = fuzz_parity.draw_case(rng, t)
    for T in os.environ.get("DBG_T", "512").split(","):
        os.environ["ORBX_OCT_THREADS"] = T.rstrip("r")
        os.environ.pop("ORBX_OCT_ROOMY", None)
        if T.endswith("r"): os.environ["ORBX_OCT_ROOMY"] = "1"
        try:
            ex = X.ORBextractor(c["nf"], c["sf"], c["nlevels"], c["ini"], c["mn"], max_width=c["cols"], max_height=c["rows"])
        except X.OrbxError:
            print(t, "rejected"); break
        o, want = oracle_run(c["img"], c["nf"], c["lap"], c["nlevels"], c["sf"], c["ini"], c["mn"])
        mono, k, d, lvl = ex(c["img"], None, c["lap"])
        bad = [l for l in range(c["nlevels"]) if lvl[l].tobytes() != o.level_keypoints(l).tobytes()]
        tabs = X.compute_tables(c["nf"], c["sf"], c["nlevels"])
        print("case", t, "T", T, "%dx%d nf=%d levels=%d" % (c["cols"], c["rows"], c["nf"], c["nlevels"]), "bad levels", bad, "quotas", tabs["features_per_level"].tolist(),
              "cands", [len(o.candidates(l)) for l in range(c["nlevels"])], "kept", [len(o.level_keypoints(l)) for l in range(c["nlevels"])])
        for l in bad:
            g, w = lvl[l], o.level_keypoints(l)
            gs = set(map(tuple, np.stack([g["x"], g["y"]], 1).tolist())) if len(g) else set()
            ws = set(map(tuple, np.stack([w["x"], w["y"]], 1).tolist())) if len(w) else set()
            print("   level", l, "gpu n", len(g), "oracle n", len(w), "common", len(gs & ws), "gpu-only", sorted(gs - ws)[:6], "oracle-only", sorted(ws - gs)[:6])
