cd "$(dirname "$0")/../.."
for m in 125m 350m; do for l in 0 2; do LTR_LANES=$l python tests/diag/lanes_lab.py $m /tmp/lanes_${m}_$l.npz; done
python - <<PY
import numpy as np
a, b = np.load("/tmp/lanes_${m}_0.npz"), np.load("/tmp/lanes_${m}_2.npz")
print("$m max |score(lanes) - score(one lane)| per k:", {k: float(np.abs(a[k] - b[k]).max()) for k in a.files})
PY
done
