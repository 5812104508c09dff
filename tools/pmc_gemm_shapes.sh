#!/bin/bash
# PMC passes (counters only) over the mid-size / ragged / decode GEMM shapes, product library through the torch ops:
#   FETCH / WRITE bytes against the algorithmic bytes, L2 hit rate, instruction mix, LDS conflicts.   tools/pmc_gemm_shapes.sh <outdir>
OUT=${1:-gpurun_out/pmc_gemm_shapes}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/pmc_gemm_driver.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, qutlass_amd as q
from qutlass_amd.utils import to_blocked
dev = torch.device("cuda", 0); torch.manual_seed(0)
def had(n):
    h = torch.ones(1, 1)
    while h.shape[0] < n: h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(dev)
h32, h16 = had(32), had(16)
alpha = torch.tensor([1.0], device=dev); gs = torch.tensor([1.0], device=dev)
def mx(m, k):
    x = torch.randn(m, k, dtype=torch.bfloat16, device=dev) * 25
    xq, xs = q.fusedQuantizeMx(x, h32, method="abs_max"); return xq, to_blocked(xs)
def nv(m, k):
    x = torch.randn(m, k, dtype=torch.bfloat16, device=dev) * 25
    xq, xs = q.fusedQuantizeNv(x, h16, gs); return xq, to_blocked(xs)
shapes = [(512, 4096, 4096), (1024, 4096, 4096), (2048, 4096, 4096), (4096, 5120, 4096), (16, 4096, 4096), (256, 14336, 4096), (2048, 4096, 8192)]
ops = []
for (m, n, k) in shapes:
    a, sa = mx(m, k); b, sb = mx(n, k)
    ops.append(lambda a=a, b=b, sa=sa, sb=sb: q.matmul_mxf4_bf16_tn(a, b, sa, sb, alpha))
for (m, n, k) in [(1024, 4096, 4096), (2048, 4096, 4096)]:
    a, sa = nv(m, k); b, sb = nv(n, k)
    ops.append(lambda a=a, b=b, sa=sa, sb=sb: q.matmul_nvf4_bf16_tn(a, b, sa, sb, alpha))
torch.cuda.synchronize()
for _ in range(5):
    for f in ops: f()
torch.cuda.synchronize()
PY
run() { rocprofv3 --pmc $2 -d $R/$OUT/$1 -o p -- python /tmp/pmc_gemm_driver.py > $R/$OUT/$1.log 2>&1; }
run tcc1 "FETCH_SIZE"
run tcc2 "WRITE_SIZE"
run tcc3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run sq3 "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
cd $R
python - <<PY
import sqlite3, glob, collections
per = collections.defaultdict(dict)   # (dispatch order index within the run, kernel name) -> {counter: value}
for db in sorted(glob.glob("$OUT/*/p_results.db")):
    c = sqlite3.connect(db)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        kn = "kernel_name" if "kernel_name" in cols else "name"
        gcols = [x for x in cols if "grid" in x.lower()]
        gsel = gcols[0] if gcols else "0"
        q = f"select dispatch_id, {kn}, {gsel}, counter_name, value from counters_collection order by dispatch_id"
        seq = collections.OrderedDict()
        for did, name, gx, ctr, v in c.execute(q):
            if not any(t in name for t in ("qamd::gemm", "skinny", "splitk")): continue
            ent = seq.setdefault(did, (name, gx, {}))
            ent[2][ctr] = ent[2].get(ctr, 0) + v
        items = list(seq.values())
        n = len(items) // 5          # dispatches per iteration of the driver's op list
        for idx, (name, gx, d) in enumerate(items[-n:]):
            per[(idx, name[:100], gx)].update(d)
    except Exception as ex:
        print("ERR", db, repr(ex), cols if "cols" in dir() else None)
with open("$OUT/summary.txt", "w") as f:
    for (idx, name, gx), d in sorted(per.items()):
        f.write("#%d %s grid=%s\n" % (idx, name, gx))
        f.write("    " + "  ".join("%s=%.0f" % (k, d[k]) for k in sorted(d)) + "\n")
print(open("$OUT/summary.txt").read()[:9000])
PY
rm -rf $OUT/tcc1 $OUT/tcc2 $OUT/tcc3 $OUT/sq2 $OUT/sq3
