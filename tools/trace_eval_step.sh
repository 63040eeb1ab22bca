#!/bin/bash
# Run ON THE GPU BOX: kernel table of the RotatE evaluation step (2 048 users x 45 539 items -> top-10) of the product path.
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_eval
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/eval_one.py <<PY
import sys, os, time
sys.path.insert(0, "$ROOT")
import numpy as np, torch
from tests.helpers import FakeDataset, make_config, random_tables
from tests.test_gpu_kge import build
dev = torch.device("cuda", 0)
model, dim, nU, nI, nE, nR, B, k = "RotatE", 200, 6041, 45539, 46000, 13, 2048, 10
m = build(model, random_tables(model, nU, nE, nR, dim, seed=0), dim, nU, nI, nE, nR)
rng = np.random.default_rng(0)
users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
hist = (torch.from_numpy(rng.integers(0, B, 40 * B)).to(dev), torch.from_numpy(rng.integers(1, nI, 40 * B)).to(dev))
with torch.no_grad():
    for _ in range(5): m.full_sort_topk({"user_id": users}, k, hist)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.full_sort_topk({"user_id": users}, k, hist)
    torch.cuda.synchronize(); print("ms per step", (time.perf_counter() - t0) / 20 * 1e3)
PY
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 /tmp/eval_one.py > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = 25
for r in rows[:16]:
    print(f'{int(r["Calls"])/steps:6.2f} {int(r["TotalDurationNs"])/steps/1e3:8.1f} us  {r["Name"][:100]}')
PY
find $OUT -name "*kernel_trace.csv" -delete
