"""NumPy stand-in for cmtf_pls_amd.backend.HipBackend -- TEST INFRASTRUCTURE ONLY.

It lets the engine's control flow (component / iteration / block loops, sharded reductions, the
R2X / R2Y identities, the normal-equation solve) run on CPU tensors, so that it can be compared
with the oracle and exercised with world_size-2 gloo jobs without a GPU.  It is never importable
from the product package and implements each kernel's contract (include/cmtfpls.h) in plain NumPy.
"""
import numpy as np
import torch


def _np(t):
    return None if t is None else t.numpy()


class NumpyBackend:
    name = "numpy-test"
    device = torch.device("cpu")

    def empty(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype)

    zeros = empty

    def colstats(self, X2):
        x = _np(X2).astype(np.float64)
        obs = ~np.isnan(x)
        return torch.from_numpy(np.where(obs, x, 0.0).sum(0)), torch.from_numpy(obs.sum(0).astype(np.float64))

    def center(self, X2, mean, want_rowcnt):
        x = _np(X2)
        x[...] = (x.astype(np.float64) - _np(mean)).astype(x.dtype)
        obs = ~np.isnan(x)
        rowcnt = torch.from_numpy(obs.sum(1).astype(np.float64)) if want_rowcnt else None
        return rowcnt, torch.tensor([float(np.sum(np.where(obs, x, 0.0).astype(np.float64) ** 2))], dtype=torch.float64)

    def mode0_contract(self, X2, u, masked, out=None):
        x = _np(X2).astype(np.float64)
        if masked:
            x = np.where(np.isnan(x), 0.0, x)
        z = torch.from_numpy(x.T @ _np(u))
        if out is not None:
            out.copy_(z)
            return out
        return z

    n_partials = 4          # partial rows of score_gram (the HIP backend has one per workgroup)

    def mode0_contract_yq(self, X2, Y, q, masked, out):
        if X2.shape[1] % 2 == 1:
            return None        # stands for "shape outside the fused form": the engine must fall back
        return self.mode0_contract(X2, Y @ q, masked, out=out)

    def deflate_contract_yq(self, X2, A, B, t, wA, wB, Y, q, masked, out):
        if X2.shape[1] % 2 == 1:
            return None
        ssq = self.deflate(X2, A, B, t, wA, wB)
        self.mode0_contract(X2, Y @ q, masked, out=out)
        return ssq

    def score_gram(self, X2, A, B, wA, wB, rowcnt, out, Y, qpart):
        if Y.shape[1] > 64:
            return None
        self.score(X2, A, B, wA, wB, rowcnt, out)
        M = Y.shape[1]
        parts = qpart.reshape(-1)[: self.n_partials * M].view(self.n_partials, M)
        parts.zero_()
        for g in range(self.n_partials):                      # rows dealt to the partials round-robin
            parts[g] = Y[g::self.n_partials].t() @ out[g::self.n_partials]
        return out

    def q_update(self, q, qpart=None, normalize=True, G=None, q_prev=None, du2=None, nparts=None):
        M = q.numel()
        if qpart is not None:
            n = nparts or self.n_partials
            q.copy_(qpart.reshape(-1)[: n * M].view(n, M).sum(0))
        if normalize:
            q /= torch.linalg.norm(q)
        if G is not None:
            d = q - q_prev
            du2[0] = float(d @ G @ d)

    def xcov(self, X2, Y, masked, out=None, mixed=False):
        x = _np(X2).astype(np.float64)
        if masked:
            x = np.where(np.isnan(x), 0.0, x)
        S = torch.from_numpy(_np(Y).T @ x)
        if out is not None:
            out.copy_(S)
            return out
        return S

    def xcov_iterate(self, S, A, B, q_cur, Z, wA, wB, info, n_squarings, q_new, G, du2, first):
        if first:
            self.mode0_contract(S, q_cur, False, out=Z)
        self.rank1(Z, A, B, wA, wB, info=info, n_squarings=n_squarings)
        self.score(S, A, B, wA, wB, None, q_new)
        self.q_update(q_new, None, True, G, q_cur, du2)

    def xcov_blocks_plan(self, blocks, M, q_cur, tq, q_new, G, status):
        if M > 64 or any(b["order"] not in (2, 3) for b in blocks):
            return None
        Tq = tq.view(len(blocks), M)

        def enqueue(n_squarings, first):
            for i, b in enumerate(blocks):
                if first:
                    self.mode0_contract(b["S"], q_cur, False, out=b["Z"])
                    if b["colcnt"] is not None:
                        self.colscale(b["Z"], b["colcnt"], b["n_samples"])
                if b["order"] == 3:
                    self.rank1(b["Z"], b["A"], b["B"], b["wA"], b["wB"], info=status[1 + 2 * i: 3 + 2 * i], n_squarings=n_squarings[i])
                else:
                    b["wB"].copy_(b["Z"] / torch.linalg.norm(b["Z"]))
                self.score(b["S2"] if b["S2"] is not None else b["S"], b["A"], b["B"], b["wA"], b["wB"], None, Tq[i])
            q_new.copy_(Tq.mean(dim=0) if len(blocks) > 1 else Tq[0])
            self.q_update(q_new, None, True, G, q_cur, status[0:1])
        return enqueue

    # -- the uncentred ("raw") cross-covariance fit and the fused masked deflation: contracts of include/cmtfpls.h ----------------
    def axpy_scalar(self, y, a, x=None):
        y -= float(a[0]) * (x if x is not None else 1.0)
        return y

    def total(self, v):
        return torch.tensor([float(v.sum())], dtype=torch.float64)

    def recon_r2(self, X2, T, WA, WB, mean):
        x = _np(X2).astype(np.float64) - (_np(mean) if mean is not None else 0.0)
        W = (_np(WA)[:, None, :] * _np(WB)[None, :, :]).reshape(x.shape[1], -1)
        xhat = _np(T) @ W.T
        ok = np.isfinite(x)
        return torch.tensor([float(((xhat - x)[ok] ** 2).sum()), float((x[ok] ** 2).sum())], dtype=torch.float64)

    def xcov_ssq(self, X2, Y, mean, out):
        x = _np(X2).astype(np.float64)
        out.copy_(torch.from_numpy(_np(Y).T @ x))
        return out, torch.tensor([float(((x - _np(mean)) ** 2).sum())], dtype=torch.float64)

    def xcov_stats(self, X2, Y, out):
        if Y.shape[1] > 64:
            return None
        x = _np(X2).astype(np.float64)
        out.copy_(torch.from_numpy(_np(Y).T @ x))
        return out, torch.from_numpy(np.concatenate([x.sum(axis=0), (x * x).sum(axis=0)]))

    def xcov_deflate(self, X2, A, B, Y, t, wA, wB, out):
        if X2.shape[1] % 4 != 0 or Y.shape[1] > 64:
            return None        # (the HIP kernel takes whole 4-element vectors only: the engine must deflate, then rebuild)
        ssq = self.deflate(X2, A, B, t, wA, wB)
        self.xcov(X2, Y, True, out=out)
        return ssq

    def status_snapshot(self, status, slot, slots=None):
        return status.clone().numpy()

    def status_wait(self, token):
        return token

    def kr_axpy(self, v, A, B, WA, WB, k, coef):
        W = (_np(WA)[:, None, :k] * _np(WB)[None, :, :k]).reshape(A * B, k)
        vv = _np(v)
        vv -= W @ _np(coef)[:k]
        return v

    def s_downdate(self, S, A, B, ya, wA, wB, q, v):
        S -= torch.outer(ya.reshape(-1), torch.from_numpy(self._w(wA, wB))) + torch.outer(q, v)

    def quadform(self, G, q, q_old, out):
        d = q - q_old
        out[0] = float(d @ G @ d)
        return out

    def mttkrp(self, X2, A, B, WA, WB, out, mixed=False):
        W = (_np(WA)[:, None, :] * _np(WB)[None, :, :]).reshape(A * B, -1)
        out.copy_(torch.from_numpy(_np(X2).astype(np.float64) @ W))
        return out

    def colscale(self, Z, colcnt, n_samples):
        z, c = _np(Z), _np(colcnt)
        with np.errstate(all="ignore"):
            z[...] = np.where(c > 0, z / c * n_samples, 0.0)

    rank1_squarings = 30

    def rank1(self, Z, A, B, wA, wB, info=None, n_squarings=None):
        if info is not None:
            info[0], info[1] = 1.0, 0.0
        U, S, Vt = np.linalg.svd(_np(Z).reshape(A, B), full_matrices=False)
        u, v = U[:, 0], Vt[0]
        if v[np.argmax(np.abs(v))] < 0:
            u, v = -u, -v
        wA.copy_(torch.from_numpy(u.copy()))
        wB.copy_(torch.from_numpy(v.copy()))

    def rank1_tensor(self, Z, dims, tol, factors, info=None, n_squarings=None):
        from oracle import rank1_factors   # test infrastructure may use the oracle
        for m, f in enumerate(rank1_factors(_np(Z).reshape(dims), tol)):
            factors[m, : len(f)] = torch.from_numpy(np.asarray(f).copy())
        if info is not None:
            info[0], info[1] = 1.0, 0.0

    def kron(self, a, b, out):
        out.copy_(torch.from_numpy(np.kron(_np(a), _np(b))))
        return out

    def normalize(self, v):
        v /= torch.linalg.norm(v)

    def _w(self, wA, wB):
        return np.kron(_np(wA), _np(wB))

    def score(self, X2, A, B, wA, wB, rowcnt, out):
        x = _np(X2).astype(np.float64)
        w = self._w(wA, wB)
        if rowcnt is not None:
            with np.errstate(all="ignore"):
                t = np.where(np.isnan(x), 0.0, x) @ w / _np(rowcnt) * w.shape[0]
        else:
            t = x @ w
        out.copy_(torch.from_numpy(t))
        return out

    def score_s(self, S, A, B, wA, wB, out):
        return self.score(S, A, B, wA, wB, None, out)

    def score_contract(self, X2, A, B, wA, wB, shift, t, Z, sub_own=None, add_other=None, alpha=1.0, csum=None):
        if X2.shape[1] % 2 == 1:
            return None        # stands for "row outside the registers of one workgroup": the engine must make the two passes
        x = _np(X2).astype(np.float64)
        s = x @ self._w(wA, wB) - (float(shift[0]) if shift is not None else 0.0)
        if sub_own is not None:
            s = s - _np(sub_own)
        t.copy_(torch.from_numpy(s))
        c = alpha * (s + (_np(add_other) if add_other is not None else 0.0))
        Z.copy_(torch.from_numpy(x.T @ c))
        if csum is not None:
            csum[0] = float(np.sum(c))
        return Z

    def deflate(self, X2, A, B, t, wA, wB):
        x = _np(X2)
        x[...] = (x.astype(np.float64) - np.outer(_np(t), self._w(wA, wB))).astype(x.dtype)
        return torch.tensor([float(np.nansum(x.astype(np.float64) ** 2))], dtype=torch.float64)

    def score_deflate(self, X2, A, B, wA, wB, rowcnt, out):
        self.score(X2, A, B, wA, wB, rowcnt, out)
        return self.deflate(X2, A, B, out, wA, wB)

    def gram_tn(self, A, B, out=None):
        if A.dim() == 1:
            A = A.unsqueeze(1)
        if B.dim() == 1:
            B = B.unsqueeze(1)
        C = (A.t() @ B).contiguous()
        if out is not None:
            out.copy_(C.view(out.shape))
            return out
        return C

    def rowdot(self, Y, q, u_out, u_old, du2=None):
        u = Y @ q
        if u_old is not None:
            val = float(((u_old - u) ** 2).sum())
            if du2 is None:
                du2 = torch.zeros(1, dtype=torch.float64)
            du2[0] = val
        u_out.copy_(u)
        return du2 if u_old is not None else None

    def normal_solve(self, G, g, out=None):
        Gh, gh = _np(G), _np(g).reshape(-1)
        d = np.where(np.diag(Gh) > 0, 1.0 / np.sqrt(np.where(np.diag(Gh) > 0, np.diag(Gh), 1.0)), 0.0)
        z = np.linalg.lstsq(Gh * d[:, None] * d[None, :], gh * d, rcond=None)[0]
        b = torch.from_numpy(np.ascontiguousarray(z * d))
        if out is not None:
            out.copy_(b)
            return out
        return b

    def unit_upper_solve_rows(self, M, U, shift=None, nan_flag=None):
        R = M.shape[1]
        tri = np.eye(R) + np.triu(_np(U), 1)
        m = _np(M).copy()
        if nan_flag is not None and np.isnan(m).any():
            nan_flag.fill_(1)
        if shift is not None:
            m = m - _np(shift)[None, :]
        M.copy_(torch.from_numpy(np.ascontiguousarray(np.linalg.solve(tri.T, m.T).T)))
        return M

    def kr_gram_row(self, L, a, g, first):
        row = torch.from_numpy(_np(L)[:, :a].T @ _np(L)[:, a])
        if first:
            g[:a] = row
        else:
            g[:a] *= row
        return g

    def kr_gram(self, L, G, first):
        g = torch.from_numpy(_np(L).T @ _np(L)).reshape(G.shape)
        if first:
            G.copy_(g)
        else:
            G.mul_(g)
        return G

    def khatri_rao(self, Am, Bm):
        R = Am.shape[1]
        return torch.from_numpy(np.ascontiguousarray((_np(Am)[:, None, :] * _np(Bm)[None, :, :]).reshape(-1, R)))

    def scores_mean(self, Ts, out):
        out.copy_(torch.from_numpy(np.average(_np(Ts), axis=0)))
        return out

    def y_deflate(self, Y, T, ncols, b, q):
        Y -= torch.outer(T[:, :ncols] @ b, q)
        return torch.tensor([float((Y ** 2).sum())], dtype=torch.float64)
