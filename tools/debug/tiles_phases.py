"""Per-phase cycles of the tiled rollout kernel (build with GPMPC_EXTRA_DEFS=-DGPMPC_TILES_PHASES)."""
import os, sys, ctypes as C
os.environ["GPMPC_ROLLOUT_TILES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sampling_gpmpc_amd as sg
import bench_configs as bc
pname, Ns, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sys.argv = ["x"]
bc.run(pname, Ns, H, False, 5)
raw = C.CDLL(sg._lib.LIB_PATH)
buf = (C.c_double * 4096)()
raw.gpmpc_debug_read_tiles(buf)
ph = np.array(buf)[63 * 64: 63 * 64 + 8]
names = ["input + A (real block)", "B + C (rhs of hall. rows)", "D (solve; streamed rows in ii)", "E + F (gram, exchange)", "G (roots, sample)", "H (append)", "D: resident rows (regime ii)", "-"]
tot = ph.sum()
for n, v in zip(names, ph):
    print(f"  {n:28s} {v / H:10.0f} cycles/step  {100 * v / max(tot, 1):5.1f} %")
print(f"  total {tot / H:.0f} cycles/step, {tot:.0f} per rollout")
sub = np.array(buf)[63 * 64 + 8: 63 * 64 + 24]
snames = ["A1 state / feedback / input", "A2 exp + axis products", "A3 grid entries -> LDS", "A4 real-block Gram (MFMA)", "B1 kernel entries", "B2 Kronecker correction",
          "B3/C lane-map conversion", "H1 record / labels / C -> LDS", "H2 new rows' tile stores", "H3 diagonal tiles", "H4 row reload into AGPRs"]
for n, v in zip(snames, sub):
    print(f"      {n:32s} {v / H:10.0f} cycles/step")
