"""Developer tool: bench.py against another build of the library (EQUSS_LIB=<path to a libequss_hip.so>), for
A/B runs inside one GPU call (boxes differ by several percent).  Arguments are bench.py's."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
if os.environ.get("EQUSS_LIB"):
    pkg.native.LIB_PATH = os.environ["EQUSS_LIB"]
import bench
bench.main()
