"""How many rows of a grouped block's per-neighbour tensors are real?  ball_query pads a neighbourhood with its first
hit, so a query with count points in its ball contributes max(count, 1) distinct rows out of nsample = 32.  numpy, CPU:
the bench input (x ~ N(0,1), condition U[-1,1]^3) and a sphere of radius 0.5 at t = 500 / 200 / 50 / 0 of the T = 1000
schedule (x_t = sqrt(abar) x_0 + sqrt(1 - abar) eps) with a partial sphere as the condition.
    python -m tools.lab.ball_duplicates > profiles/r4_ball_duplicates.txt"""
import numpy as np

from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch


def fps(p, m):
    idx, d = [0], np.full(len(p), 1e10)
    for _ in range(m - 1):
        d = np.minimum(d, ((p - p[idx[-1]]) ** 2).sum(1))
        idx.append(int(d.argmax()))
    return np.array(idx)


def counts(q, s, r, ns=32):
    return np.minimum((((q[:, None, :] - s[None, :, :]) ** 2).sum(-1) < r * r).sum(1), ns)


def report(tag, xc, cc):
    lx, lc = [xc], [cc]
    for m in (1024, 256, 64, 16):
        lx.append(lx[-1][fps(lx[-1], m)])
        lc.append(lc[-1][fps(lc[-1], m)])
    print(tag)
    for name, rad, pairs in (("SA", [.1, .2, .4, .8], [(lx[l + 1], lx[l]) for l in range(4)]),
                             ("encoder feature transfer", [.1, .2, .4, .8], [(lx[l + 1], lc[l + 1]) for l in range(4)]),
                             ("decoder feature transfer", [.1, .2, .4, .8, 1.6], [(lx[l], lc[l]) for l in range(5)])):
        for l, (q, s) in enumerate(pairs):
            c = counts(q, s, rad[l])
            print("    %-26s %d: mean count %5.1f  max %2d  <= 1 point %3.0f%%  distinct rows %3.0f%%" % (
                name, l, c.mean(), c.max(), 100 * (c <= 1).mean(), 100 * np.maximum(c, 1).mean() / 32))


def main():
    x, cond, _ = synthetic_batch(1, seed=0)
    report("bench input (x ~ N(0,1), condition U[-1,1]^3)", x[0].numpy(), cond[0, :, :3].numpy())
    rng = np.random.default_rng(0)
    sph = rng.normal(size=(2048, 3))
    sph = 0.5 * sph / np.linalg.norm(sph, axis=1, keepdims=True)
    part = sph[sph[:, 0] > -0.1][:1536]
    part = np.concatenate([part, part * np.array([1, 1, -1.])])[:3072]
    for t, ab in ((500, 0.078), (200, 0.66), (50, 0.97), (0, 1.0)):
        xt = np.sqrt(ab) * sph + np.sqrt(1 - ab) * rng.normal(size=sph.shape)
        report("sphere r = 0.5 at t = %d (alpha_bar %.2f), condition = partial sphere" % (t, ab),
               xt.astype(np.float32), part.astype(np.float32))


if __name__ == "__main__":
    main()
