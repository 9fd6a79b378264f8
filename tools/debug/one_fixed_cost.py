import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["x", "--sustained"]
import bench_configs as bc
for h in (2, 3, 4, 6, 10):
    bc.run("params_pendulum1D_samples", 1024, h, False, 20)
