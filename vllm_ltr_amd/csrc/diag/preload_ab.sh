# A/B of the kernarg-preload build flag (build.py PRELOAD): the shipped library against one built with LTR_NO_KERNARG_PRELOAD=1
#   LTR_NO_KERNARG_PRELOAD=1 build -> copy libltr_hip.so to csrc/lab_old/, then build normally;  bash diag/preload_ab.sh
L=$GRAFT_REPO_ROOT/vllm_ltr_amd/csrc/lab_old/libltr_hip.so
O=gpurun_out/preload; mkdir -p $O
for i in 1 2; do
 echo "== without preload"; LTR_LIB=$L python tests/diag/small_call_profile.py 1 2 4 16 64 256 2>&1 | grep "^k=" | cut -c1-60
 echo "== preload (shipped)"; python tests/diag/small_call_profile.py 1 2 4 16 64 256 2>&1 | grep "^k=" | cut -c1-60
done
cat > /tmp/dump.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from util import bench_lengths, synthetic_batch
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.scorer import HipOPTScorer
res = []
for spec in (OPTSpec.opt_125m(), OPTSpec.opt_350m()):
    sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
    for k in (1, 3, 16, 64, 700):
        lens = bench_lengths(max(k, 256), seed=k)[:k]
        ids, cu = synthetic_batch(spec, lens.tolist(), 1)
        out = torch.empty(k, device="cuda:0")
        sc.score_device(torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda(), cu, out=out)
        res.append(out.cpu().numpy())
np.save(sys.argv[1], np.concatenate(res))
PY
LTR_LIB=$L python /tmp/dump.py $O/old.npy 2>&1 | grep -v amdgpu.ids | tail -3; python /tmp/dump.py $O/new.npy 2>&1 | grep -v amdgpu.ids | tail -3
python -c "
import numpy as np; a=np.load('$O/old.npy'); b=np.load('$O/new.npy'); print('scores of both families, 1 ... 700 requests:', a.size, 'bit-identical' if (a.view(np.uint32)==b.view(np.uint32)).all() else ('max|d| %g' % np.abs(a-b).max()))"
