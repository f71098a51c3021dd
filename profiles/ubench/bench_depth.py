import os, sys, runpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tfrec_amd import api
api.FIFO_DEPTH = int(os.environ["PY_FIFO_DEPTH"])
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
