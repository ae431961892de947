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
    def __init__(self, model, m, n, k, bc, h, omega, mu, device, part, group=None, flags=0):
        if model not in (E.MODEL_MF_NN_TV, E.MODEL_MF_NN_CURV):
            raise E.CpEngineError("cpengine:unsupported row slabs are implemented for the mf_nn models")
        import os
        # NN-TV runs the fused kernel (one gather per iteration); NN-curvature the two field passes of cp_mfnn_curv.cuh
        # (CP_TUNE_CURV_SPLIT=1: the split passes of round 1), two gathers per iteration either way
        self.fused = (model == E.MODEL_MF_NN_TV)
        self.curv_fused = (not self.fused) and os.environ.get("CP_TUNE_CURV_SPLIT", "0") in ("", "0")
        self.model, self.bc, self.h, self.omega, self.mu, self.flags = model, bc, h, omega, mu, flags
        self.part, self.m, self.n, self.k = part, m, n, k
        self.device = device
        self.dev = torch.device("cuda", device)
        self.comm = SlabComm(part.rank, part.world, group)
        self.eng = E.Engine(model, m, part.nl, k=k, bc=bc, h=h, omega=omega, mu=mu, device=device, n_global=n,
                            j_offset=part.j0, flags=flags)
        # The engine and the collectives share ONE real (non-default) stream, made the current torch stream of this thread:
        # on the legacy default stream every hand-over between the engine's kernels and RCCL goes through the implicit
        # null-stream synchronisation, which measured 60 us per iteration (1.62 -> 1.56 ms at 4096 x 512 x 16).
        # CP_MG_DEFAULT_STREAM=1 keeps whatever stream is current.
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
        # message buffers (doubles; the fp32 mode packs two floats per double): state selections and the vector
        # selections of the power iteration
        sels = (E.SEL_X_U, E.SEL_P_Y12, E.SEL_P_DY, E.SEL_VEC_X_U, E.SEL_VEC_P_DY) if self.fused else \
            (E.SEL_X_U, E.SEL_P_Y2, E.SEL_VEC_X_U, E.SEL_VEC_P_Y2)
        for sel in sels:
            ln = self.eng.halo_len(sel)
            self.hbuf[sel] = [torch.zeros(ln, **f64) for _ in range(4)]   # [send a, recv a, send b, recv b]
        if self.fused:
            self.len_xu, self.len_py = self.eng.halo_len(E.SEL_X_U), self.eng.halo_len(E.SEL_P_Y12)
            # one message per neighbour: [u | y1, y2] packed back to back in the storage type
            esz = 4 if (flags & E.FLAG_FP32) else 8
            l7 = (7 * k * m * esz + 7) // 8
            ldy = (2 * k * m * esz + 7) // 8
            self.s_left = torch.zeros(l7, **f64)     # -> left neighbour's right halo
            self.r_right = torch.zeros(l7, **f64)    # <- right neighbour's first column
            self.s_right = torch.zeros(ldy, **f64)   # -> right neighbour's left halo
            self.r_left = torch.zeros(ldy, **f64)    # <- left neighbour's last column

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

    def norm_estimate(self, x0=None):
        """NNE/matrix_norm.m:15-48 with halo exchanges around A and A' and a scalar all-reduce per step; the start vector
        is the engine's deterministic one (global indices, so it equals the single-handle start)"""
        if x0 is not None:
            raise E.CpEngineError("cpengine:unsupported a caller-supplied start vector on row slabs")
        ss = torch.tensor([self.eng.mg_power_start()], dtype=torch.float64)
        ss = ss.to(self.dev) if not self.comm.stage_cpu else ss
        self.comm.all_reduce_sum(ss)
        scale = 1.0 / math.sqrt(float(ss.item()))
        e = []
        i = 0
        while i < 2 or abs(e[i - 1] - e[i - 2]) > 1e-6 * e[i - 1]:
            i += 1
            # the vectors of the power iteration are doubles in every storage mode: CP_SEL_VEC_* selections
            if self.fused:
                self._send_first_to_left(E.BUF_X1, E.SEL_VEC_X_U)
            else:
                self._exchange_both(E.BUF_X1, E.SEL_VEC_X_U)
            self.eng.mg_stage(E.STAGE_APPLY_A, 0, self.red.data_ptr(), scale=scale)
            if self.fused:
                self._send_last_to_right(E.BUF_P1, E.SEL_VEC_P_DY)
            else:
                self._exchange_both(E.BUF_P1, E.SEL_VEC_P_Y2)
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
        if self.fused or self.curv_fused:
            # first dual step alone (y_1 from xbar_0 = x_0), W_1, the mean of the first un-projected u
            eng.mg_stage(E.STAGE_DUAL0, 0, self.red.data_ptr())
            self.comm.all_gather(self.gathered, self.red)
            eng.mg_stage(E.STAGE_EIG0, 0, self.gathered.data_ptr(), world=w)
            self.exchange_p(E.BUF_P1)
        self.it_n = 0

    def iterate_steps(self, iters, check_stop=0, tol=0.0):
        """`iters` more CP iterations; per iteration: fused pass, all-gather of the reduced partials, eigensolve, and
        one halo message per neighbour that is in flight during the eigensolve"""
        eng, w = self.eng, self.part.world
        if self.curv_fused:
            # NN-curvature, two field passes: P(n) -> gather -> [halo(u_n) | eigensolve] -> D(n) -> gather (3 scalars) ->
            # [halo(y2_{n+1}) | fin]. D(n) needs the halo columns of u_n (5-point Laplacian of ubar_n); P(n+1) those of
            # y2_{n+1}. As in the fused NN-TV path the messages are packed before the gather and issued after it.
            for n in range(self.it_n + 1, self.it_n + iters + 1):
                xb, pb = E.BUF_X0 + (n & 1), E.BUF_P0 + ((n + 1) & 1)
                eng.mg_stage(E.STAGE_CURV_P, n, self.red.data_ptr())
                self._exchange_both_pack(xb, E.SEL_X_U)
                self.comm.all_gather(self.gathered, self.red)
                tok = self._exchange_both_begin(E.SEL_X_U)
                eng.mg_stage(E.STAGE_EFUSED, n, self.gathered.data_ptr(), world=w)
                self._exchange_both_end(tok, xb, E.SEL_X_U)
                eng.mg_stage(E.STAGE_CURV_Y2, n, self.red.data_ptr())
                self._exchange_both_pack(pb, E.SEL_P_Y2)
                self.comm.all_gather(self.gathered, self.red)
                tok = self._exchange_both_begin(E.SEL_P_Y2)
                eng.mg_stage(E.STAGE_CURV_FIN, n, self.gathered.data_ptr(), world=w, check_stop=check_stop, tol=tol)
                self._exchange_both_end(tok, pb, E.SEL_P_Y2)
            self.it_n += iters
            return
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

    def iterate_end(self, iterations=None):
        self.iterations = self.it_n if iterations is None else iterations
        self.eng.mg_end(self.iterations)

    def solve(self, theta, tau, sigma, max_iter, tol, chunk=32):
        """chambolle_pock on the slabs (PDO/chambolle_pock.m:54-228): history row 0, iterations with the stop rule
        evaluated on the device (every rank runs the identical single-workgroup kernels on the gathered partials, so all
        ranks see the same flag), polled every `chunk` iterations as cp_solve does. Returns the engine's result record;
        iteration counts and histories equal the single-handle solve."""
        eng, w = self.eng, self.part.world
        if not (self.fused or self.curv_fused):
            raise E.CpEngineError("cpengine:unsupported MultiGpuEngine.solve runs the fused passes (unset CP_TUNE_CURV_SPLIT)")
        self.iterate_begin_row0(theta, tau, sigma, max_iter)
        done, stopped, stop_iter = 0, 0, -1
        while done < max_iter and not stopped:
            cnt = min(chunk, max_iter - done)
            self.iterate_steps(cnt, check_stop=1, tol=tol)
            done += cnt
            stopped, stop_iter = eng.mg_stop_state()
        self.iterate_end(stop_iter if stopped else max_iter)
        return eng.result()

    def iterate_begin_row0(self, theta, tau, sigma, max_iters):
        """iterate_begin preceded by history row 0 = primal_objective(x0), dual_objective(y0) (:89-106)"""
        eng, w = self.eng, self.part.world
        self.exchange_x(E.BUF_X0)
        self.exchange_p(E.BUF_P0)
        eng.mg_begin(theta, tau, sigma, max_iters)
        eng.mg_stage(E.STAGE_ROW0_DUAL, 0, self.red.data_ptr())
        self.comm.all_gather(self.gathered, self.red)
        eng.mg_stage(E.STAGE_ROW0_EIG, 0, self.gathered.data_ptr(), world=w)
        eng.mg_stage(E.STAGE_ROW0_PRIMAL, 0, self.red.data_ptr())
        self.comm.all_gather(self.gathered, self.red)
        eng.mg_stage(E.STAGE_ROW0_FIN, 0, self.gathered.data_ptr(), world=w)
        if self.fused or self.curv_fused:
            eng.mg_stage(E.STAGE_DUAL0, 0, self.red.data_ptr())
            self.comm.all_gather(self.gathered, self.red)
            eng.mg_stage(E.STAGE_EIG0, 0, self.gathered.data_ptr(), world=w)
            self.exchange_p(E.BUF_P1)
        self.it_n = 0

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

    def result(self):
        return self.eng.result()

    # ---- multi-level driver support --------------------------------------------------------------------
    def prolong_from(self, coarse, factor=1.0):
        """x of this level <- kron(x_coarse, ones(2)) cropped (NNE/mf_nn_tv...:113-119) on the owned columns; `coarse` is a
        single-GPU Engine holding the whole coarse level (replicated on every rank) or a MultiGpuEngine whose slab
        boundaries are half of this one's (slab.hierarchical_partitions) -- no communication either way. The dual is
        reset to zero (:122); the halo columns follow with the next linearize()."""
        ceng = coarse.eng if isinstance(coarse, MultiGpuEngine) else coarse
        self.eng.prolong_from(ceng, factor)

    def downsample_to(self, coarse):
        """images are replicated on every rank: the pyramid (:75-76) is computed locally"""
        ceng = coarse.eng if isinstance(coarse, MultiGpuEngine) else coarse
        self.eng.downsample_to(ceng)

    def primal_full(self):
        """the whole primal vector in the reference's layout on EVERY rank (gathers the slabs; for results, not for the
        iteration)"""
        nf = self.eng.nx // (self.m * self.part.nl)
        mine = (self.part.j0, self.part.j1, self.eng.primal())
        parts = [None] * self.part.world
        dist.all_gather_object(parts, mine, group=self.comm.group)
        X = np.empty((nf, self.n, self.m))
        for (j0, j1, xl) in parts:
            X[:, j0:j1, :] = xl.reshape(nf, j1 - j0, self.m)
        return X.reshape(-1)

    def dual(self):
        return self.eng.dual()

    def history(self):
        return self.eng.history()

    def close(self):
        self.eng.close()
