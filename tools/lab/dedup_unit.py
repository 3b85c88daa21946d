"""pdr_dedup_plan + pdr_gather_add_tiles + per-query pdr_gather_add + pdr_weighted_moments against the whole
pdr_gather_add: per-cloud moment sums must agree (lab check)."""
import torch
from point_diffusion_refinement_amd import _lib


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    B, m, K, Cout, n_src, relu_col0 = 3, 1024, 32, 96, 700, 64
    ld = Cout
    U = torch.randn(B * n_src + 1, ld, device=dev, generator=g)
    V2 = torch.randn(B * m, 2 * ld, device=dev, generator=g)
    counts = torch.randint(0, 3, (B, m), device=dev, dtype=torch.int32, generator=g)
    counts[torch.rand(B, m, device=dev, generator=g) < 0.1] = 7
    first = torch.randint(0, n_src, (B, m, 1), device=dev, dtype=torch.int32, generator=g)
    rest = torch.randint(0, n_src, (B, m, K), device=dev, dtype=torch.int32, generator=g)
    slot = torch.arange(K, device=dev)[None, None, :]
    idx = torch.where(slot < counts[:, :, None], rest, first.expand(B, m, K)).contiguous()
    idx[:, :, 0] = first[:, :, 0]
    st = torch.cuda.current_stream().cuda_stream
    tpb = m * K // 128
    full = torch.empty(B * tpb, Cout, 2, device=dev)
    args = (U.data_ptr(), ld, n_src, V2.data_ptr(), V2.data_ptr() + 4 * ld, 2 * ld)
    _lib.check(lib.pdr_gather_add(*args, idx.data_ptr(), counts.data_ptr(), None, None, None, None, B, m * K, K, Cout,
                                  None, ld, full.data_ptr(), relu_col0, 0, -1, st), "ga")
    idx0 = torch.empty(B, m, dtype=torch.int32, device=dev)
    row_w = torch.empty(B * m, device=dev)
    tv = torch.empty(B * tpb, dtype=torch.uint8, device=dev)
    tl = torch.empty(B * tpb, dtype=torch.int32, device=dev)
    nt = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(lib.pdr_dedup_plan(idx.data_ptr(), counts.data_ptr(), B, m, K, idx0.data_ptr(), row_w.data_ptr(),
                                  tv.data_ptr(), tl.data_ptr(), nt.data_ptr(), st), "plan")
    torch.cuda.synchronize()
    want_valid = (counts.view(-1, 4) > 1).any(1)
    print("plan: valid tiles %d of %d (want %d), list ok %s, idx0 ok %s, row_w ok %s" % (
        int(nt), B * tpb, int(want_valid.sum()), bool(torch.equal(tl[:int(nt)].long(), want_valid.nonzero()[:, 0])),
        bool(torch.equal(idx0, idx[:, :, 0])),
        bool(torch.equal(row_w.view(-1, 4), (~want_valid)[:, None].float().expand(-1, 4) * K))))
    tpbd = (m + 127) // 128
    ptpb = tpb + tpbd
    part = torch.full((B * ptpb, Cout, 2), float("nan"), device=dev)
    _lib.check(lib.pdr_gather_add_tiles(*args, idx.data_ptr(), counts.data_ptr(), None, None, None, None, B, m * K, K,
                                        Cout, None, ld, part.data_ptr(), relu_col0, 0, -1, tv.data_ptr(), ptpb, st),
               "ga_tiles")
    Yd = torch.empty(B * m, ld, device=dev)
    _lib.check(lib.pdr_gather_add(*args, idx0.data_ptr(), counts.data_ptr(), None, None, None, None, B, m, 1, Cout,
                                  Yd.data_ptr(), ld, None, relu_col0, 0, -1, st), "ga1")
    _lib.check(lib.pdr_weighted_moments(Yd.data_ptr(), ld, B, m, Cout, relu_col0, row_w.data_ptr(), part.data_ptr(),
                                        ptpb, tpb, tv.data_ptr(), st), "wm")
    torch.cuda.synchronize()
    a = full.view(B, tpb, Cout, 2).double().sum(1)
    b = part.view(B, ptpb, Cout, 2).double().sum(1)
    print("nan rows in dedup partial:", int(torch.isnan(part).any(2).any(1).sum()))
    print("per-cloud moment sums: max rel diff %.3e" % float(((a - b).abs() / (a.abs() + 1)).max()))
    # pieces: valid tiles equal?
    pv = part.view(B, ptpb, Cout, 2)[:, :tpb].reshape(B * tpb, Cout, 2)
    v = tv.bool()
    print("valid tiles' rows equal:", bool(torch.equal(pv[v], full[v])), " skipped rows zero:", bool((pv[~v] == 0).all()))
    want_deg = (full[~v].double().sum(0))
    got_deg = part.view(B, ptpb, Cout, 2)[:, tpb:].double().sum((0, 1))
    print("skipped tiles' moments: max rel diff %.3e" % float(((want_deg - got_deg).abs() / (want_deg.abs() + 1)).max()))


if __name__ == "__main__":
    main()
