import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ.get("AGT_LIB", "libagt_hip_knobs.so"))
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
bench.main()
