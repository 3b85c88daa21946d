# Session of the r6_double_accumulator experiment (apply the patch, `make -C point_diffusion_refinement_amd/csrc`):
#   python -m pytest tests/test_fused_gpu.py -m gpu -q -k two_accumulator      # bit-identity with the two-workgroup form
#   bash tools/lab/ab_double_acc.sh                                             # layers alone + the step, option off / on
#   lab builds: -DPDR_LAB_TRACE (tools/lab/ws_trace.py), -DPDR_LAB_DA_NOSTORE (the multiply waves without their stores)
# Numbers: profiles/r6_second_half_ab.txt (sections r6z_da*), DESIGN.md 4.10 (3).
bash tools/lab/ab_double_acc.sh
