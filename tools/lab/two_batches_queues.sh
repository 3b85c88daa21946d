# two samplers in flight under different numbers of hardware queues (GPU_MAX_HW_QUEUES; the default is 4)
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 300 python - <<'PY' 2>&1 | grep samplers
import json
from tools.lab import two_batches as T
for form in ("once",):
    for n in (1, 2, 1, 2):
        print(json.dumps(T.run(n, 40, form)), flush=True)
PY
done
