"""Row-slab multi-GPU execution of the NN-TV hot path: one process per GPU over torch.distributed (RCCL).

Per CP iteration:
    k_mf_fused_tv on the rank's slab -> local reduction -> ONE all-gather of the reduced partials
    (20 scalars + two K x K Gram matrices) -> every rank runs the identical fixed-order sum + eigensolve
    (k_mf_efused) -> neighbour exchange of one halo column of u (x_n) and of y_{n+1}.
No ring all-reduce: the payload is 4 KB, latency-bound, and the fixed-order local sum keeps all ranks bitwise identical.

`SlabComm` is the communication layer (pure torch.distributed, works with gloo on CPU tensors -> tested without a GPU);
`MultiGpuEngine` binds it to the cp_mg_* entry points of libcpengine.so.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import engine as E


class SlabComm:
    """neighbour halo exchange + all-gather for a 1-D chain of ranks (no wrap-around)"""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.backend = dist.get_backend(group)
        self.stage_cpu = (self.backend == "gloo")

    def _wire(self, t):
        # gloo cannot move device tensors: stage through host memory (only used in tests on a one-GPU box)
        return t.cpu() if (self.stage_cpu and t.is_cuda) else t

    def exchange_begin(self, to_left=None, to_right=None, from_left=None, from_right=None):
        """starts sending `to_left` to rank-1 and `to_right` to rank+1 and receiving into `from_left` / `from_right`
        what the neighbours sent towards this rank; buffers may be None at the ends of the chain or when a direction
        is unused. Returns a token for exchange_end (the transfers overlap whatever is enqueued in between)."""
        ops, post = [], []
        if self.left is not None and to_left is not None:
            ops.append(dist.P2POp(dist.isend, self._wire(to_left), self.left, self.group))
        if self.right is not None and to_right is not None:
            ops.append(dist.P2POp(dist.isend, self._wire(to_right), self.right, self.group))
        if self.left is not None and from_left is not None:
            buf = torch.empty_like(from_left, device="cpu") if (self.stage_cpu and from_left.is_cuda) else from_left
            ops.append(dist.P2POp(dist.irecv, buf, self.left, self.group))
            post.append((from_left, buf))
        if self.right is not None and from_right is not None:
            buf = torch.empty_like(from_right, device="cpu") if (self.stage_cpu and from_right.is_cuda) else from_right
            ops.append(dist.P2POp(dist.irecv, buf, self.right, self.group))
            post.append((from_right, buf))
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, post

    def exchange_end(self, token):
        works, post = token
        for w in works:
            w.wait()          # RCCL: the current stream waits, the host does not
        for dst, buf in post:
            if dst is not buf:
                dst.copy_(buf)

    def exchange(self, to_left=None, to_right=None, from_left=None, from_right=None):
        self.exchange_end(self.exchange_begin(to_left, to_right, from_left, from_right))

    def all_gather(self, gathered, mine):
        """gathered: (world, len) tensor, row r = rank r's vector (fixed order -> deterministic reductions)"""
        if self.stage_cpu and mine.is_cuda:
            g = torch.empty(gathered.numel(), dtype=gathered.dtype)
            dist.all_gather_into_tensor(g, mine.cpu(), group=self.group)
            gathered.view(-1).copy_(g)
        else:
            dist.all_gather_into_tensor(gathered.view(-1), mine, group=self.group)

    def all_reduce_sum(self, t):
        w = self._wire(t)
        dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
        if w is not t:
            t.copy_(w)
        return t


class MultiGpuEngine:
    def __init__(self, model, m, n, k, bc, h, omega, mu, device, part, group=None):
        if model not in (E.MODEL_MF_NN_TV, E.MODEL_MF_NN_CURV):
            raise E.CpEngineError("cpengine:unsupported row slabs are implemented for the mf_nn models")
        # NN-TV runs the fused kernel (one gather per iteration); NN-curvature the split passes (two gathers)
        self.fused = (model == E.MODEL_MF_NN_TV)
        self.part, self.m, self.n, self.k = part, m, n, k
        self.dev = torch.device("cuda", device)
        self.comm = SlabComm(part.rank, part.world, group)
        self.eng = E.Engine(model, m, part.nl, k=k, bc=bc, h=h, omega=omega, mu=mu, device=device, n_global=n,
                            j_offset=part.j0)
        # The engine and the collectives share ONE real (non-default) stream, made the current torch stream of this thread:
        # on the legacy default stream every hand-over between the engine's kernels and RCCL goes through the implicit
        # null-stream synchronisation, which measured 60 us per iteration (1.62 -> 1.56 ms at 4096 x 512 x 16).
        # CP_MG_DEFAULT_STREAM=1 keeps whatever stream is current.
        import os
        if os.environ.get("CP_MG_DEFAULT_STREAM", "0") != "1" and \
                torch.cuda.current_stream(self.dev) == torch.cuda.default_stream(self.dev):
            self.stream = torch.cuda.Stream(device=self.dev)
            torch.cuda.set_stream(self.stream)
        self.eng.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        plen = int(E.lib.cp_mg_partial_len())
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.red = torch.zeros(plen, **f64)
        self.gathered = torch.zeros((part.world, plen), **f64)
        self.hbuf = {}
        sels = (E.SEL_X_U, E.SEL_P_Y12, E.SEL_P_DY) if self.fused else (E.SEL_X_U, E.SEL_P_Y2)
        for sel in sels:
            ln = self.eng.halo_len(sel)
            self.hbuf[sel] = [torch.zeros(ln, **f64) for _ in range(4)]   # [send a, recv a, send b, recv b]
        if self.fused:
            self.len_xu, self.len_py = self.eng.halo_len(E.SEL_X_U), self.eng.halo_len(E.SEL_P_Y12)
            ldy = self.eng.halo_len(E.SEL_P_DY)
            self.s_left = torch.zeros(self.len_xu + self.len_py, **f64)     # -> left neighbour's right halo
            self.r_right = torch.zeros(self.len_xu + self.len_py, **f64)    # <- right neighbour's first column
            self.s_right = torch.zeros(ldy, **f64)                          # -> right neighbour's left halo
            self.r_left = torch.zeros(ldy, **f64)                           # <- left neighbour's last column

    # ---- halo exchanges ---------------------------------------------------------------------------
    def _send_first_to_left(self, buf, sel):
        """first owned column -> left neighbour's right halo"""
        send, recv = self.hbuf[sel][0], self.hbuf[sel][1]
        c = self.comm
        if c.left is not None:
            self.eng.halo_pack(buf, sel, 0, send.data_ptr())
        c.exchange(to_left=send if c.left is not None else None, from_right=recv if c.right is not None else None)
        if c.right is not None:
            self.eng.halo_unpack(buf, sel, 1, recv.data_ptr())

    def _send_last_to_right(self, buf, sel):
        """last owned column -> right neighbour's left halo"""
        send, recv = self.hbuf[sel][0], self.hbuf[sel][1]
        c = self.comm
        if c.right is not None:
            self.eng.halo_pack(buf, sel, 1, send.data_ptr())
        c.exchange(to_right=send if c.right is not None else None, from_left=recv if c.left is not None else None)
        if c.left is not None:
            self.eng.halo_unpack(buf, sel, 0, recv.data_ptr())

    def _exchange_both(self, buf, sel):
        """first owned column -> left neighbour's right halo AND last owned column -> right neighbour's left halo"""
        sl, rr, sr, rl = self.hbuf[sel]
        c = self.comm
        if c.left is not None:
            self.eng.halo_pack(buf, sel, 0, sl.data_ptr())
        if c.right is not None:
            self.eng.halo_pack(buf, sel, 1, sr.data_ptr())
        c.exchange(to_left=sl if c.left is not None else None, to_right=sr if c.right is not None else None,
                   from_left=rl if c.left is not None else None, from_right=rr if c.right is not None else None)
        if c.right is not None:
            self.eng.halo_unpack(buf, sel, 1, rr.data_ptr())
        if c.left is not None:
            self.eng.halo_unpack(buf, sel, 0, rl.data_ptr())

    def _exchange_both_pack(self, buf, sel):
        """_exchange_both in three parts: pack, start the messages (_exchange_both_begin), wait + unpack
        (_exchange_both_end), so that the kernel enqueued between the last two hides the transfer"""
        sl, rr, sr, rl = self.hbuf[sel]
        c = self.comm
        if c.left is not None:
            self.eng.halo_pack(buf, sel, 0, sl.data_ptr())
        if c.right is not None:
            self.eng.halo_pack(buf, sel, 1, sr.data_ptr())

    def _exchange_both_begin(self, sel):
        sl, rr, sr, rl = self.hbuf[sel]
        c = self.comm
        return c.exchange_begin(to_left=sl if c.left is not None else None, to_right=sr if c.right is not None else None,
                                from_left=rl if c.left is not None else None, from_right=rr if c.right is not None else None)

    def _exchange_both_end(self, token, buf, sel):
        sl, rr, sr, rl = self.hbuf[sel]
        c = self.comm
        c.exchange_end(token)
        if c.right is not None:
            self.eng.halo_unpack(buf, sel, 1, rr.data_ptr())
        if c.left is not None:
            self.eng.halo_unpack(buf, sel, 0, rl.data_ptr())

    def exchange_x(self, buf):
        if self.fused:
            self._send_first_to_left(buf, E.SEL_X_U)          # forward differences: column j+1 only
        else:
            self._exchange_both(buf, E.SEL_X_U)               # 5-point Laplacian: columns j-1 and j+1

    def exchange_p(self, buf):
        if self.fused:
            self._send_first_to_left(buf, E.SEL_P_Y12)
            self._send_last_to_right(buf, E.SEL_P_DY)
        else:
            self._exchange_both(buf, E.SEL_P_Y2)

    def exchange_state_pack(self, xbuf, pbuf):
        """one message per neighbour and iteration: [u of x_n | y1, y2 of y_{n+1}] of the first owned column goes left,
        the Dy components of y2 of the last owned column go right"""
        c = self.comm
        self.eng.state_pack(xbuf, pbuf, self.s_left.data_ptr() if c.left is not None else 0,
                            self.s_right.data_ptr() if c.right is not None else 0)

    def exchange_state_begin(self):
        c = self.comm
        return c.exchange_begin(to_left=self.s_left if c.left is not None else None,
                                to_right=self.s_right if c.right is not None else None,
                                from_left=self.r_left if c.left is not None else None,
                                from_right=self.r_right if c.right is not None else None)

    def exchange_state_end(self, token, xbuf, pbuf):
        c = self.comm
        c.exchange_end(token)
        self.eng.state_unpack(xbuf, pbuf, self.r_right.data_ptr() if c.right is not None else 0,
                              self.r_left.data_ptr() if c.left is not None else 0)

    # ---- engine facade ----------------------------------------------------------------------------
    def set_images(self, imgs_full):
        self.eng.set_images(imgs_full)

    def set_primal(self, x0_local):
        self.eng.set_primal(x0_local)
        self.exchange_x(E.BUF_X0)

    def set_dual(self, y0_local):
        self.eng.set_dual(y0_local)

    def linearize(self):
        self.exchange_x(E.BUF_X0)
        self.eng.linearize()

    def set_nu(self, nu):
        self.eng.set_nu(nu)

    def singular_values(self, which):
        """Gram -> eig, then a second Gram in the rotated basis (cf. cp_singular_values), Gram summed over ranks"""
        def gram(V):
            g = torch.from_numpy(self.eng.mg_gram(which, V).copy())
            g = g.to(self.dev) if not self.comm.stage_cpu else g
            self.comm.all_reduce_sum(g)
            return g.cpu().numpy()
        k = self.k
        G = gram(None)
        _, V1 = np.linalg.eigh(0.5 * (G[:k, :k] + G[:k, :k].T))
        V = np.eye(16)
        V[:k, :k] = V1
        G2 = gram(V)
        lam = np.linalg.eigvalsh(0.5 * (G2[:k, :k] + G2[:k, :k].T))
        return np.sqrt(np.maximum(lam, 0.0))[::-1].copy()

    def norm_estimate(self):
        """NNE/matrix_norm.m:15-48 with halo exchanges around A and A' and a scalar all-reduce per step"""
        ss = torch.tensor([self.eng.mg_power_start()], dtype=torch.float64)
        ss = ss.to(self.dev) if not self.comm.stage_cpu else ss
        self.comm.all_reduce_sum(ss)
        scale = 1.0 / math.sqrt(float(ss.item()))
        e = []
        i = 0
        while i < 2 or abs(e[i - 1] - e[i - 2]) > 1e-6 * e[i - 1]:
            i += 1
            self.exchange_x(E.BUF_X1)
            self.eng.mg_stage(E.STAGE_APPLY_A, 0, self.red.data_ptr(), scale=scale)
            if self.fused:
                self._send_last_to_right(E.BUF_P1, E.SEL_P_DY)
            else:
                self._exchange_both(E.BUF_P1, E.SEL_P_Y2)
            self.eng.mg_stage(E.STAGE_APPLY_AT, 0, self.red.data_ptr())
            s = self.red[:1].clone()
            self.comm.all_reduce_sum(s)
            normY = math.sqrt(float(s.item()))
            scale = 1.0 / normY
            e.append(math.sqrt(normY))
            if i == 100:
                break
        return e[-1], i

    def iterate_begin(self, theta, tau, sigma, max_iters=4000):
        """start-up: halos of the canonical state, first dual step (y_1), W_1"""
        eng, w = self.eng, self.part.world
        self.exchange_x(E.BUF_X0)
        self.exchange_p(E.BUF_P0)
        eng.mg_begin(theta, tau, sigma, max_iters)       # x_old = x0 (halo columns included); history rows 0..max_iters
        if self.fused:
            eng.mg_stage(E.STAGE_DUAL0, 0, self.red.data_ptr())
            self.comm.all_gather(self.gathered, self.red)
            eng.mg_stage(E.STAGE_EIG0, 0, self.gathered.data_ptr(), world=w)
            self.exchange_p(E.BUF_P1)
        self.it_n = 0

    def iterate_steps(self, iters, check_stop=0, tol=0.0):
        """`iters` more CP iterations; per iteration: fused pass, all-gather of the reduced partials, eigensolve, and
        one halo message per neighbour that is in flight during the eigensolve"""
        eng, w = self.eng, self.part.world
        if not self.fused:
            # split passes: dual -> gather -> [halo(y2) | eig] -> primal -> gather -> [halo(u) | fin]: the halo messages
            # (both directions, the Laplacian is symmetric) need nothing from the single-workgroup kernels -- y2 is not
            # touched by the eigensolve, the stored u carries its mean offset lazily -- so they are in flight meanwhile.
            # They are issued AFTER the gather (as in the fused path): the gather is on the critical path and must not
            # queue behind a megabyte of halo data if RCCL runs both on one stream.
            for n in range(self.it_n + 1, self.it_n + iters + 1):
                pb, xb = E.BUF_P0 + (n & 1), E.BUF_X0 + (n & 1)
                eng.mg_stage(E.STAGE_SPLIT_DUAL, n, self.red.data_ptr())
                self._exchange_both_pack(pb, E.SEL_P_Y2)
                self.comm.all_gather(self.gathered, self.red)
                tok = self._exchange_both_begin(E.SEL_P_Y2)
                eng.mg_stage(E.STAGE_SPLIT_EIG, n, self.gathered.data_ptr(), world=w, check_stop=check_stop, tol=tol)
                self._exchange_both_end(tok, pb, E.SEL_P_Y2)
                eng.mg_stage(E.STAGE_SPLIT_PRIMAL, n, self.red.data_ptr())
                self._exchange_both_pack(xb, E.SEL_X_U)
                self.comm.all_gather(self.gathered, self.red)
                tok = self._exchange_both_begin(E.SEL_X_U)
                eng.mg_stage(E.STAGE_SPLIT_FIN, n, self.gathered.data_ptr(), world=w)
                self._exchange_both_end(tok, xb, E.SEL_X_U)
            self.it_n += iters
            return
        for n in range(self.it_n + 1, self.it_n + iters + 1):
            xb, pb = E.BUF_X0 + (n & 1), E.BUF_P0 + ((n + 1) & 1)
            eng.mg_stage(E.STAGE_FUSED, n, self.red.data_ptr())
            self.exchange_state_pack(xb, pb)
            self.comm.all_gather(self.gathered, self.red)
            tok = self.exchange_state_begin()
            eng.mg_stage(E.STAGE_EFUSED, n, self.gathered.data_ptr(), world=w, check_stop=check_stop, tol=tol)
            self.exchange_state_end(tok, xb, pb)
        self.it_n += iters

    def iterate_end(self):
        self.eng.mg_end(self.it_n)
        self.iterations = self.it_n

    def iterate(self, theta, tau, sigma, iters, check_stop=0, tol=0.0):
        """`iters` CP iterations from the canonical state (benchmark hook; the stop rule is off by default)"""
        self.iterate_begin(theta, tau, sigma, iters)
        self.iterate_steps(iters, check_stop, tol)
        self.iterate_end()

    def sync(self):
        self.eng.sync()

    def kernel_time_ms(self, name):
        return self.eng.kernel_time_ms(name)

    def primal(self):
        return self.eng.primal()

    def dual(self):
        return self.eng.dual()

    def history(self):
        return self.eng.history()

    def close(self):
        self.eng.close()
