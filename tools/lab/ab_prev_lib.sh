python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "attention_pool or ddpm_config" 2>&1 | tail -2
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-160
  cp point_diffusion_refinement_amd/libpdr_hip.so /tmp/new.so; cp point_diffusion_refinement_amd/libpdr_lab.so point_diffusion_refinement_amd/libpdr_hip.so
  python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-160
  cp /tmp/new.so point_diffusion_refinement_amd/libpdr_hip.so
done
