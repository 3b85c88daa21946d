"""HBM read / write / copy rates of plain torch kernels at the sizes of the narrow layers (268 MB .. 1 GB):
what a streaming kernel can reach on this chip when it only reads, only writes, or does both."""
import torch


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = torch.device("cuda:0")
    for mb in (64, 268, 537, 1074):
        n = mb * 1000 * 1000 // 4
        a = torch.empty(n, device=dev)
        b = torch.empty(n, device=dev)
        a.normal_()
        big = torch.empty(n * 3, device=dev)          # cycled through so that nothing stays in the 256 MB MALL
        w = t(lambda: (big.fill_(1.0), a.fill_(2.0))) 
        wr = (4 * n * 4) / w / 1e12
        r = t(lambda: (big.sum(), a.sum()))
        rd = (4 * n * 4) / r / 1e12
        c = t(lambda: b.copy_(a))
        cp = (2 * n * 4) / c / 1e12
        print("%5d MB: write %.2f TB/s   read %.2f TB/s   copy (r+w) %.2f TB/s" % (mb, wr, rd, cp))


if __name__ == "__main__":
    main()
