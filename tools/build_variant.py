#!/usr/bin/env python3
"""Build the kernel library with extra compiler flags into a side file, for several-library A/Bs on one box (tools/ab_lib_gemm.py, tools/ab_lib_shapes.py;
gpu_session.sh steps ablib* / abpairs with AB_OLD / AB_NEW / AB_PAIRS):

    python tools/build_variant.py build/exp/libqamd_magic.so -DQAMD_CTX_MAGIC_DECODE=1                          # the PRODUCT units + flags
    python tools/build_variant.py build/exp/lab_splitb.so --lab -DQAMD_ROUTE_LABK -DQAMD_DEEPP_SPLITB=1         # the LAB units + flags
    gpurun -- 'AB_DATA=zero python tools/ab_lib_gemm.py qutlass_amd/libqutlass_amd.so build/exp/lab_base.so build/exp/lab_splitb.so'

(build/ is git-ignored and travels to the GPU box with the snapshot.)  Product switches that exist for this: QAMD_CTX_MAGIC_DECODE, QAMD_RING_KERNARG_EARLY /
QAMD_NV_KERNARG_EARLY (per-tile MX / NVFP4 kernels).  The persistent kernels' experiments ([r5]) live in the lab copy (csrc/lab/gemm_mx_deepp_lab.hip.h: QAMD_DEEPP_SPLITB,
QAMD_DEEPP_RETIRE, QAMD_DEEPP8_RETIRE, QAMD_DEEPP_FS_IL, QAMD_FS_BURST, QAMD_FS_ABL, QAMD_DEEPP_RB2, QAMD_DEEPP_SOFF, QAMD_KERNARG_EARLY, QAMD_DEEPP_PEEL,
QAMD_DEEPP_EARLYPREP): `--lab -DQAMD_ROUTE_LABK` makes the plain matmul_mxf4_bf16_tn entry of a lab build run that copy (a `lab_base.so` built with the routing flag alone
is the control).  `AB_DATA=zero` times the schedules in cycles (no data-dependent power: tools/power_data_probe.py)."""
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
