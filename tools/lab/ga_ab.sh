python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "small_config or ddpm_config or optional" 2>&1 | tail -2
python -m tools.gather_add_bench > gpurun_out/c17_new.txt
python - <<'PY' > gpurun_out/c17_prev.txt
from point_diffusion_refinement_amd import _lib
_lib.LIB_PATH = "point_diffusion_refinement_amd/libpdr_lab.so"
import runpy, sys
sys.argv = ["x"]
runpy.run_module("tools.gather_add_bench", run_name="__main__")
PY
paste gpurun_out/c17_new.txt gpurun_out/c17_prev.txt | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,"|",$(NF-3),$(NF-2)}'
