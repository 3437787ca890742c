O=gpurun_out/r03_h; mkdir -p $O
python -m pytest tests/test_gpu_attention.py -m gpu -x -q -s > $O/t_attn.log 2>&1; grep -E "max\||passed|failed" $O/t_attn.log | tail -12
for v in 2 4 6 2 6; do
  LTR_FLAGS_LTR_ATTN="-DLTR_ATTN_VSWZ=$v" python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-strong --no-unfused --no-class-head --steady-new 0 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VSWZ=$v', round(d['ms_per_step'],2), 'attn', round(d['kernels']['attn']['ms_per_step'],3), 'gemm', round(d['kernels']['gemm']['ms_per_step'],2))"
done | tee $O/ab_vswz.txt
cd /tmp && export TMPDIR=/tmp
for v in 2 6; do
  (cd $GRAFT_REPO_ROOT && LTR_FLAGS_LTR_ATTN="-DLTR_ATTN_VSWZ=$v" python -m vllm_ltr_amd.csrc.build > /dev/null 2>&1)
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $GRAFT_REPO_ROOT/$O/pmc$v -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-unfused --no-profile-pass --no-strong --no-class-head --steady-new 0 > $GRAFT_REPO_ROOT/$O/pmc$v.log 2>&1
  python - <<PY
import sqlite3,glob
db=glob.glob("$GRAFT_REPO_ROOT/$O/pmc$v/**/*.db", recursive=True)[0]
cur=sqlite3.connect(db).cursor()
t={}
for name,c,v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    if "attn_f16s" in name: t[c]=t.get(c,0)+v
print("VSWZ=$v attn_f16s:", {k: int(x) for k,x in t.items()}, "conflict frac", t.get("SQ_LDS_BANK_CONFLICT",0)/max(t.get("SQ_LDS_IDX_ACTIVE",1),1))
PY
done | tee -a $GRAFT_REPO_ROOT/$O/ab_vswz.txt
find $GRAFT_REPO_ROOT/$O -name "*.db" -delete
