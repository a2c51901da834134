"""per-instance LDS footprint of the benchmark worlds and the dimensions behind it (runs without a GPU).
usage: python tools/lds_dims.py [config ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["RKFD_DEVMODEL_DUMP"] = "1"
B = importlib.import_module("roki-fd_amd.binding")
sc = importlib.import_module("roki-fd_amd.scenarios")
for w in sys.argv[1:] or ["config2", "config3", "config4", "config4v", "config5"]:
    s = sc.CONFIGS[w](4)
    n = B.lib().rkfdLdsBytesFor(s["world"].model, s["max_rigid"])
    chunks = -(-n // 1280)           # the hardware hands LDS out in 1280-byte pieces, 128 per CU (tools/ubench/residency.hip)
    print("%-9s %6d B = %2d pieces -> %2d resident per CU (LDS); 12 at most from the registers" % (w, n, chunks, 128 // chunks), flush=True)
