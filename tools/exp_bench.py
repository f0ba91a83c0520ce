"""bench.py on a timing-only experiment build (VIPNERF_HIP_LIB=...libvipnerf_hip_<variant>.so): the product-build guard off.  The line it
prints is NOT a result -- experiment builds leave work out or race on purpose; for A/B timings of tools/build_variant.sh variants only.
    VIPNERF_HIP_LIB=vip-nerf_amd/lib/libvipnerf_hip_exp0.so VIPNERF_EXP_OVERLAP=1 python tools/exp_bench.py --precision bf16 --no-configs2 --no-sizes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from vipnerf_hip import _lib
_lib.require_product_build = lambda who: None
import bench
sys.argv = ['bench.py'] + sys.argv[1:]
bench.main()
