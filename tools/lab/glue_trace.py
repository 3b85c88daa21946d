"""Which Python call sites issue the small non-native kernels (memcpy, fill, elementwise, cat) of one cached
reverse step?  torch.profiler with stacks over ONE eager step; prints kernel name -> count -> top user frames.
    python -m tools.lab.glue_trace"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_sampler  # noqa: E402
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    sampler, _ = build_sampler(dev, use_graph=False)
    x_T, cond, label = synthetic_batch(32, seed=0, device=dev)
    sampler.begin((32, 2048, 3), cond, label, x_T=x_T)
    sampler.advance(2)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        sampler.advance(1)
        torch.cuda.synchronize()
    by_op = collections.defaultdict(lambda: [0, collections.Counter()])
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::"):
            if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::contiguous", "aten::clone",
                           "aten::_to_copy", "aten::index_select", "aten::mul", "aten::add", "aten::sub", "aten::div",
                           "aten::empty", "aten::zeros", "aten::pad", "aten::constant_pad_nd", "aten::linear",
                           "aten::reciprocal", "aten::sum", "aten::sigmoid", "aten::sin", "aten::cos", "aten::randn_like"):
                frames = [f for f in (ev.stack or []) if "point_diffusion_refinement_amd" in f or "bench.py" in f]
                key = frames[0].split("point_diffusion_refinement_amd/")[-1] if frames else "?"
                by_op[ev.name][0] += 1
                by_op[ev.name][1][key] += 1
    for name, (n, sites) in sorted(by_op.items(), key=lambda kv: -kv[1][0]):
        print("%-24s %4d" % (name, n))
        for site, c in sites.most_common(6):
            print("      %3d  %s" % (c, site[:150]))
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))


if __name__ == "__main__":
    main()
