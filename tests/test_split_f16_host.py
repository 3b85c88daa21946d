"""Host side of the opt-in split-f16 arithmetic (no GPU): the packed weight image and its contract."""
import pytest
import torch


def test_split_f16_weight_image_layout_and_contract():
    """Host side of the opt-in split-f16 arithmetic (fused_network.pack_f16x3, layout contract in include/pdr_hip.h):
    per column block and 32-channel chunk one [hi | lo] x [TN columns][32 k] block of halves whose 16-byte granule g of
    column n sits at position g ^ ((n >> 2) & 3); hi + lo reproduces the weight to 2^-22 (2^-25 absolute for small
    values); segments are cut into chunks separately, the last chunk of each zero-padded; weights beyond the f16 range
    are refused."""
    from point_diffusion_refinement_amd.pointnet2.fused_network import pack_f16x3
    g = torch.Generator().manual_seed(3)
    for Cout, segs, TN in ((128, (128,), 128), (140, (70, 41), 128), (64, (96,), 64), (40, (33, 3), 64)):
        Cin = sum(segs)
        Wt = torch.randn(Cin, Cout + 3, generator=g) * 0.1
        Wt[0, 0], Wt[1, 1] = 3e-6, 1e-9                       # subnormal hi / lo halves
        img, nch = pack_f16x3(Wt, Cout, segs, TN)
        chunks, k = [], 0
        for C in segs:
            chunks += [(k + ks, min(32, C - ks)) for ks in range(0, C, 32)]
            k += C
        ncb = (Cout + TN - 1) // TN
        assert nch == len(chunks) and img.numel() == ncb * nch * 2 * TN * 32
        blocks = img.view(torch.float16).view(ncb, nch, 2, TN, 4, 8)
        W = Wt[:, :Cout]
        for cb in range(ncb):
            for ci, (k0, km) in enumerate(chunks):
                for n in range(0, min(TN, Cout - cb * TN), 7):
                    sw = (n >> 2) & 3
                    row = torch.stack([torch.cat([blocks[cb, ci, h, n, gq ^ sw] for gq in range(4)]) for h in (0, 1)])
                    want = W[k0:k0 + km, cb * TN + n]
                    hi, lo = row[0, :km].float(), row[1, :km].float()
                    assert torch.equal(hi, want.to(torch.float16).float())
                    err = (hi.double() + lo.double() - want.double()).abs()
                    assert bool((err <= torch.maximum(want.abs().double() * 2.0 ** -22, torch.tensor(2.0 ** -25))).all())
                    assert bool((row[:, km:] == 0).all())              # zero padding of a partial chunk
    Wbad = torch.randn(64, 64, generator=g)
    Wbad[5, 7] = 7e4
    with pytest.raises(ValueError):
        pack_f16x3(Wbad, 64, (64,), 64)
