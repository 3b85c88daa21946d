# every launch walking its row tiles against the direction of the launch that wrote its main source
# (fused_network.ZIGZAG_WALK, pdr_layer_in_t.walk_reverse) vs the default forward walk: step, whole evaluation, split-f16
AB_STEPS=100 bash tools/lab/ab_opts.sh - ZIGZAG_WALK=1
AB_STEPS=100 bash tools/lab/ab_opts.sh - ZIGZAG_WALK=1
echo "== whole form"; AB_STEPS=30 AB_ARGS="--neighbourhoods whole" bash tools/lab/ab_opts.sh - ZIGZAG_WALK=1
