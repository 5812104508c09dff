#!/usr/bin/env python3
"""Build the PRODUCT translation units of the kernel library with extra compiler flags into a side file, for two-library A/Bs on one box
(tools/ab_lib_gemm.py, tools/ab_lib_shapes.py; gpu_session.sh steps ablib* with AB_OLD / AB_NEW):

    python tools/build_variant.py build/exp/libqamd_magic.so -DQAMD_CTX_MAGIC_DECODE=1
    gpurun -- 'AB_OLD=qutlass_amd/libqutlass_amd.so AB_NEW=build/exp/libqamd_magic.so bash tools/gpu_session.sh <name> ablibmx'

(build/ is git-ignored and travels to the GPU box with the snapshot.)  Compile-time switches that exist for this: QAMD_CTX_MAGIC_DECODE (per-tile MX / NVFP4 kernels
decode their tile without integer divisions, prepared at the end of round 4), QAMD_RING_KERNARG_EARLY / QAMD_NV_KERNARG_EARLY (one scalar-load round for the kernel
arguments in the per-tile MX kernels / the NVFP4 kernels, likewise prepared and unmeasured), QAMD_DEEPP_SOFF, QAMD_KERNARG_EARLY, QAMD_DEEPP_PEEL, QAMD_DEEPP_EARLYPREP (gemm_mx_deepp.hip.h)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qutlass_amd import build  # noqa: E402


def main():
    if len(sys.argv) < 2 or sys.argv[1].startswith("-"):
        sys.exit(__doc__)
    out = os.path.abspath(sys.argv[1])
    os.makedirs(os.path.dirname(out), exist_ok=True)
    flags = [a for a in sys.argv[2:] if a != "--lab"]
    if "--lab" in sys.argv:   # the LAB library with extra flags (stage traces need its TRACE instantiations): QAMD_LAB_LIB=<out> python tools/final_stage_contention.py
        units = [int(u) for u in os.environ.get("QAMD_UNITS", "").split(",") if u] or build.UNITS_BENCH
        build._compile_units(out, units, ["-DQAMD_BENCH=1"] + flags, verbose=True)
    else:
        build._compile_units(out, build.UNITS, flags, verbose=True)
    print("built", out)


if __name__ == "__main__":
    main()
