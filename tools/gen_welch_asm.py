#!/usr/bin/env python3
"""Generator (and CPU emulator) of the hand-allocated Welch kernel `mdsp_welch_w64_asm` (round 4).

Why a generator: the one-wavefront-per-transform factorisation (4096 = 64 x 64, ONE exchange, no barrier; csrc/fft_w64.h has the algebra and
tests/cpu_harness/fft_emul.cpp the host proof) only pays with TWO waves per SIMD -- a single wave issues a packed instruction every 5.7 clocks,
two every 4.75 (profiles/r02t_valu_rate.txt; variant 40 measured 1.65 ms against 1.36) -- and two waves per SIMD means <= 256 registers per
wave for 64 complex points (128) + 64 power accumulators + 28 twiddle values + the butterflies' temporaries.  hipcc needs 256 + 129 spilled
(variant 41: 2.17 ms).  Here every register is assigned by this script: values are allocated from a pool of 80 register pairs and freed at
their last use, which the script knows because it emits the dataflow itself.

What it writes:  dsp.jl_amd/csrc/welch_w64_asm.s  (assembled by build.py with clang -x assembler, linked by ld.lld, loaded with
hipModuleLoadData by csrc/welch_w64.h (included by spectral.hip)).  `--check` runs the emitted instruction list through a lane-level emulator of the ~15 opcodes it
uses (numpy, one wavefront) for two consecutive units and compares the accumulators with numpy's FFT of the windowed frame pairs.

Kernel contract (welch_run_w64asm in csrc/welch_w64.h fills the arguments):
    struct W64AsmArgs { const float* s; float* part; const float* winpairs; const float* tw; int64 lds_, units, run_len, nch; int nflush, pad; }
    grid (G, nch), 512 threads = 8 independent waves; wave w of workgroup b is slot 8 b + w and owns units [slot run_len, (slot+1) run_len) of
    its channel (a unit = two frames = 4096 new samples; every unit handed to this kernel has BOTH frames -- the odd last frame of a channel
    goes through welch_half3_kernel); every 128 units (and at the end) the 64 Float32 sums per lane are stored as one row of
    part[((slot nch + ch) nflush + f) 4096 + bin] -- a wave with U units writes exactly ceil(U / 128) rows; the reduction skips the others.
    LDS: window pairs (w[p], w[p + 2048]) 16 KiB + 8 x 16.5 KiB exchange buffers.
"""
import math
import sys
from collections import deque

import numpy as np

N = 4096
HALF = N // 2
XROW = 33                      # elements per row of the exchange buffer (fft_w64.h XP64_ROW)
XBUF_BYTES = 2 * 32 * XROW * 8
WIN_BYTES = HALF * 8
LDS_BYTES = WIN_BYTES + 8 * XBUF_BYTES
FLUSH = 128


def slot64(k):
    return (k >> 3) + 8 * (k & 7)


# ------------------------------------------------------------------------------------------------------------------ instruction records
# Operands:  ('v', p)      a 64-bit VGPR pair; p < 256 physical, p >= VBASE virtual (mapped by the allocator after scheduling)
#            ('h', p, k)   half k (0 lo, 1 hi) of pair p as a 32-bit register
#            ('r', n)      a physical 32-bit VGPR (accumulators, addresses)
#            ('s', p)      an SGPR pair (constants)
VBASE = 1000


class Ins:
    __slots__ = ("op", "d", "s", "mods", "imm", "text", "idx")

    def __init__(self, op, d=None, s=(), mods=None, imm=0, text=""):
        self.op, self.d, self.s, self.mods, self.imm, self.text, self.idx = op, d, tuple(s), mods or {}, imm, text, -1

    def operands(self):
        return ([self.d] if self.d is not None else []) + list(self.s)


def keys_of(o):
    """32-bit register units an operand covers (scheduling / liveness granularity)."""
    if o[0] == "v":
        return [("p", o[1], 0), ("p", o[1], 1)]
    if o[0] == "h":
        return [("p", o[1], o[2])]
    if o[0] == "r":
        return [("r", o[1])]
    return []


def modtext(mods, nsrc):
    out = []
    for key in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"):
        if key in mods:
            v = list(mods[key])[:nsrc]
            dflt = [1] * nsrc if key == "op_sel_hi" else [0] * nsrc
            if v != dflt:
                out.append(f"{key}:[{','.join(str(x) for x in v)}]")
    return (" " + " ".join(out)) if out else ""


def render(ins, pmap):
    """Assembly text of an instruction once its virtual pairs are mapped (pmap: virtual pair -> physical pair)."""
    def P(p):
        return pmap[p] if p >= VBASE else p

    def R(o):
        if o[0] == "v":
            return f"v[{P(o[1])}:{P(o[1]) + 1}]"
        if o[0] == "h":
            return f"v{P(o[1]) + o[2]}"
        if o[0] == "r":
            return f"v{o[1]}"
        return f"s[{o[1]}:{o[1] + 1}]"

    op = ins.op
    if op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"):
        return f"{op} {R(ins.d)}, " + ", ".join(R(x) for x in ins.s) + modtext(ins.mods, len(ins.s))
    if op == "v_fma_f32":
        return f"v_fma_f32 {R(ins.d)}, {R(ins.s[0])}, {R(ins.s[1])}, {R(ins.s[2])}"
    if op == "ds_read_b64":
        return f"ds_read_b64 {R(ins.d)}, {R(ins.s[0])} offset:{ins.imm}"
    if op == "ds_write_b64":
        return f"ds_write_b64 {R(ins.s[0])}, {R(ins.s[1])} offset:{ins.imm}"
    if op == "buffer_load_dword":
        k, imm = divmod(ins.imm, 4096)
        so = "0" if k == 0 else f"s{Gen.S_K4 + k - 1}"
        return f"buffer_load_dword {R(ins.d)}, {R(ins.s[0])}, s[{Gen.S_RS}:{Gen.S_RS + 3}], {so} offen offset:{imm}"
    if op == "v_permlane32_swap":
        return f"v_permlane32_swap_b32_e32 {R(ins.s[0])}, {R(ins.s[1])}"
    if op == "s_waitcnt_lgkm":
        return f"s_waitcnt lgkmcnt({ins.imm})"
    if op == "s_waitcnt_vm":
        return f"s_waitcnt vmcnt({ins.imm})"
    if op == "s_nop":
        return f"s_nop {ins.imm}"
    if op == "comment":
        return ins.text
    raise RuntimeError(op)


class Val:
    """A complex value in VGPR pair `p` (virtual until allocation); `.lo` / `.hi` are its 32-bit halves as operands."""
    __slots__ = ("p",)

    def __init__(self, p):
        self.p = p

    @property
    def lo(self):
        return ("h", self.p, 0)

    @property
    def hi(self):
        return ("h", self.p, 1)


class Gen:
    ACC0 = 4
    TW0 = 68
    POOL0 = 96
    NPOOL = 80
    V_OFF, V_WIN, V_XW, V_XR = 0, 1, 2, 3
    # SGPR map
    S_RS = 24      # s[24:27] sample descriptor
    S_RP = 28      # s[28:31] partial-row descriptor
    S_K4 = 32      # s32..s36 = 4096 k, k = 1..5
    S_HH = 38      # (h, h)
    S_PM = 40      # (1, -1)
    S_W = 42       # W64 roots, pairs

    W_EXPS = [1, 2, 3, 4, 5, 6, 7, 9, 10, 12, 14, 15, 18, 20, 21, 25, 28, 30, 35, 36, 42, 49]

    def __init__(self):
        self.ins = []
        self.nvirt = 0
        self.wexp = {m: self.S_W + 2 * i for i, m in enumerate(self.W_EXPS)}
        self.maxlive = 0
        self.stalls = 0

    # ---- emission (virtual registers; kill() is a no-op: liveness is computed after scheduling)
    def emit(self, op, d=None, s=(), mods=None, imm=0, text=""):
        self.ins.append(Ins(op, d, s, mods, imm, text))

    def comment(self, t):
        self.ins.append(Ins("comment", text="; " + t))

    def alloc(self):
        self.nvirt += 1
        return Val(VBASE + 2 * self.nvirt)

    def kill(self, *vals):
        pass

    def _src(self, x):
        return ("v", x.p) if isinstance(x, Val) else x

    def pk(self, op, srcs, mods=None, dst=None):
        d = dst or self.alloc()
        ss = [self._src(x) for x in srcs]
        n = len(ss)
        mods = dict(mods or {})
        for key, dflt in (("op_sel", 0), ("op_sel_hi", 1), ("neg_lo", 0), ("neg_hi", 0)):
            v = list(mods.get(key, [dflt] * n))
            v += [dflt] * (n - len(v))
            mods[key] = v
        self.emit(op, ("v", d.p), ss, mods)
        return d

    def add(self, a, b):
        return self.pk("v_pk_add_f32", [a, b])

    def sub(self, a, b):
        return self.pk("v_pk_add_f32", [a, b], {"neg_lo": [0, 1], "neg_hi": [0, 1]})

    def sub_ib(self, a, b):      # a - i b = (a.x + b.y, a.y - b.x)
        return self.pk("v_pk_add_f32", [a, b], {"op_sel": [0, 1], "op_sel_hi": [1, 0], "neg_hi": [0, 1]})

    def add_ib(self, a, b):      # a + i b = (a.x - b.y, a.y + b.x)
        return self.pk("v_pk_add_f32", [a, b], {"op_sel": [0, 1], "op_sel_hi": [1, 0], "neg_lo": [0, 1]})

    def axpy(self, u, e):        # e + h u
        return self.pk("v_pk_fma_f32", [u, ("s", self.S_HH), e])

    def axmy(self, u, e):        # e - h u
        return self.pk("v_pk_fma_f32", [u, ("s", self.S_HH), e], {"neg_lo": [1, 0, 0], "neg_hi": [1, 0, 0]})

    def cmul(self, a, w):        # complex product a w (w: Val or ('s', pair)); two instructions, the first result is reused in place
        t = self.pk("v_pk_mul_f32", [a, w], {"op_sel": [1, 1], "op_sel_hi": [1, 0]})                      # (a.y w.y, a.y w.x)
        return self.pk("v_pk_fma_f32", [a, w, t], {"op_sel_hi": [0, 1, 1], "neg_lo": [0, 0, 1]}, dst=t)      # (a.x w.x - t.x, a.x w.y + t.y)

    def mul_w64(self, a, m):
        """a W64^m (forward root), m = n1 k1 < 64."""
        if m == 0:
            return a
        if m == 16:              # -i a = (a.y, -a.x)
            return self.pk("v_pk_mul_f32", [a, ("s", self.S_PM)], {"op_sel": [1, 0], "op_sel_hi": [0, 1]})
        if m == 8:               # h (a - i a)
            t = self.sub_ib(a, a)
            return self.pk("v_pk_mul_f32", [t, ("s", self.S_HH)], dst=t)
        if m == 24:              # -h (a + i a)
            t = self.add_ib(a, a)
            return self.pk("v_pk_mul_f32", [t, ("s", self.S_HH)], {"neg_lo": [1, 0], "neg_hi": [1, 0]}, dst=t)
        return self.cmul(a, ("s", self.wexp[m]))

    # ---- butterflies (forward)
    def bfly4(self, a0, a1, a2, a3):
        t0 = self.add(a0, a2)
        t2 = self.add(a1, a3)
        t1 = self.sub(a0, a2)
        d = self.sub(a1, a3)
        x0 = self.add(t0, t2)
        x2 = self.sub(t0, t2)
        x1 = self.sub_ib(t1, d)
        x3 = self.add_ib(t1, d)
        return x0, x1, x2, x3

    def bfly8_tail(self, e, o):
        """e[0..3], o[0..3] (the two radix-4 halves) -> natural-order outputs"""
        out = [None] * 8
        out[0] = self.add(e[0], o[0])
        out[4] = self.sub(e[0], o[0])
        u1 = self.sub_ib(o[1], o[1])
        u3 = self.add_ib(o[3], o[3])
        out[2] = self.sub_ib(e[2], o[2])
        out[6] = self.add_ib(e[2], o[2])
        out[1] = self.axpy(u1, e[1])
        out[5] = self.axmy(u1, e[1])
        out[3] = self.axmy(u3, e[3])
        out[7] = self.axpy(u3, e[3])
        return out

    def bfly8(self, v):
        e = self.bfly4(v[0], v[2], v[4], v[6])
        o = self.bfly4(v[1], v[3], v[5], v[7])
        return self.bfly8_tail(e, o)

    def bfly8_sd(self, S, D):
        e0 = self.add(S[0], S[2])
        e2 = self.sub(S[0], S[2])
        o0 = self.add(S[1], S[3])
        o2 = self.sub(S[1], S[3])
        e1 = self.sub_ib(D[0], D[2])
        e3 = self.add_ib(D[0], D[2])
        o1 = self.sub_ib(D[1], D[3])
        o3 = self.add_ib(D[1], D[3])
        return self.bfly8_tail([e0, e1, e2, e3], [o0, o1, o2, o3])

    def bfly64_tail(self, v):
        for k1 in range(8):
            u = [v[j + 8 * k1] for j in range(8)]
            u = [u[0]] + [self.mul_w64(u[j], (j * k1) & 63) for j in range(1, 8)]
            out = self.bfly8(u)
            for k2 in range(8):
                v[k2 + 8 * k1] = out[k2]

    # ---- memory
    def ds_read(self, addr_vgpr, offset):
        d = self.alloc()
        self.emit("ds_read_b64", ("v", d.p), [("r", addr_vgpr)], imm=offset)
        return d

    def ds_write(self, addr_vgpr, offset, src):
        self.emit("ds_write_b64", None, [("r", addr_vgpr), ("v", src.p)], imm=offset)

    def buffer_load(self, dst, off_bytes):
        self.emit("buffer_load_dword", dst, [("r", self.V_OFF)], imm=off_bytes)

    # ---- one unit -------------------------------------------------------------------------------------------------
    def in_layout(self):
        """Fixed home of a unit's operands: XP[e'] (lo = H0, hi = H2), e' = 0..31, then HHP[n1][0], HHP[n1][1] (H1 at j = (0,1), (2,3))."""
        xp = [Val(self.POOL0 + 2 * e) for e in range(32)]
        hh = [[Val(self.POOL0 + 64 + 2 * (2 * n1 + q)) for q in range(2)] for n1 in range(8)]
        return xp, hh

    def emit_loads(self):
        xp, hh = self.in_layout()
        self.comment("the unit's three half-frames: 96 x 256-byte loads straight into the first layer's operand registers")
        for n1 in range(8):
            for j in range(4):
                e = n1 + 8 * j
                self.buffer_load(xp[e].lo, 256 * e)
                self.buffer_load(xp[e].hi, 2 * HALF * 4 + 256 * e)
            for q in range(2):
                self.buffer_load(hh[n1][q].lo, HALF * 4 + 256 * (n1 + 8 * (2 * q)))
                self.buffer_load(hh[n1][q].hi, HALF * 4 + 256 * (n1 + 8 * (2 * q + 1)))

    def emit_unit(self):
        xp, hh = self.in_layout()
        v = [None] * 64
        self.comment("pass A, first layer (window folded in) + first radix-8 layer")
        for n1 in range(8):
            wp = [self.ds_read(self.V_WIN, 512 * (n1 + 8 * j)) for j in range(4)]
            S, D = [], []
            for j in range(4):
                e = n1 + 8 * j
                h = j & 1
                T = self.pk("v_pk_mul_f32", [xp[e], wp[j]])
                s_ = self.pk("v_pk_fma_f32", [hh[n1][j >> 1], wp[j], T], {"op_sel": [h, 1, 0], "op_sel_hi": [h, 0, 1]})
                d_ = self.pk("v_pk_fma_f32", [hh[n1][j >> 1], wp[j], T],
                             {"op_sel": [h, 1, 0], "op_sel_hi": [h, 0, 1], "neg_lo": [1, 0, 0], "neg_hi": [0, 0, 1]}, dst=T)
                S.append(s_)
                D.append(d_)
            o = self.bfly8_sd(S, D)
            for k1 in range(8):
                v[n1 + 8 * k1] = o[k1]
        self.comment("pass A, second radix-8 layer (W64 roots from SGPR pairs); each butterfly's eight results are four pairs (ke, ke + 32) of the half")
        self.comment("exchange: swapped at once, and the lower halves' rows go to LDS at once (round 0 of the 64 x 64 transposition)")
        m = [None] * 64
        for k1 in range(8):
            u = [v[j + 8 * k1] for j in range(8)]
            u = [u[0]] + [self.mul_w64(u[j], (j * k1) & 63) for j in range(1, 8)]
            out = self.bfly8(u)               # out[k2] = Y_t[k1 + 8 k2]
            for k2 in range(4):
                lo_, hi_ = out[k2], out[k2 + 4]
                for half in ("lo", "hi"):
                    self.emit("v_permlane32_swap", None, [getattr(lo_, half), getattr(hi_, half)])
                m[k1 + 8 * k2] = lo_
                m[k1 + 8 * (k2 + 4)] = hi_
            for k2 in range(4):
                self.ds_write(self.V_XW, 8 * (k1 + 8 * k2), m[k1 + 8 * k2])
        nv = [None] * 64
        order = sorted(range(32), key=lambda T: (T & 7, T >> 3))     # the operands of pass B's first groups first
        for T in order:
            nv[T] = self.ds_read(self.V_XR, 8 * XROW * T)
        for r in range(32):
            self.ds_write(self.V_XW, 8 * r, m[32 + r])
        for T in order:
            nv[32 + T] = self.ds_read(self.V_XR, 8 * XROW * T)
        v = nv
        self.comment("pass B: two-level twiddles W^{8 lane t2} in front of, W^{lane t1} behind the first radix-8 layer; the products on round 0's")
        self.comment("operands (t2 < 4) come first: they cover round 1's trip through LDS")
        tw = {}
        for t2 in (1, 2, 3, 4, 5, 6, 7):
            for t1 in range(8):
                tw[(t1, t2)] = self.cmul(v[t1 + 8 * t2], Val(self.TW0 + 2 * (t2 - 1)))
        for t1 in range(8):
            q = [v[t1]] + [tw[(t1, t2)] for t2 in range(1, 8)]
            out = self.bfly8(q)
            for k1 in range(8):
                v[t1 + 8 * k1] = out[k1] if t1 == 0 else self.cmul(out[k1], Val(self.TW0 + 14 + 2 * (t1 - 1)))
        self.bfly64_tail(v)           # v[slot64(kt)] = X[lane + 64 kt]
        self.comment("power: acc[s] += re^2 + im^2")
        for s in range(64):
            a = ("r", self.ACC0 + s)
            self.emit("v_fma_f32", a, [v[s].lo, v[s].lo, a])
            self.emit("v_fma_f32", a, [v[s].hi, v[s].hi, a])

    # ---- scheduling ---------------------------------------------------------------------------------------------------
    # A wave's packed result is usable ~16 clocks after issue and, with two waves per SIMD taking turns, the wave issues every ~9.5 clocks: an
    # instruction should not sit closer than LAT issue slots behind the producers of its operands (measured: the same instructions in emission
    # order -- 200 back-to-back dependent pairs per unit -- ran at 7.1 clocks per instruction and SIMD instead of 4.75).
    LAT_VALU = 4
    LAT_LDS = 14

    def schedule(self, window=28):
        ins = [i for i in self.ins if i.op != "comment"]
        n = len(ins)
        for k, i in enumerate(ins):
            i.idx = k
        preds = [set() for _ in range(n)]
        chain = [None] * n
        last_w, readers, last_mem = {}, {}, None
        for k, i in enumerate(ins):
            rd = [key for o in i.s for key in keys_of(o)]
            wr = [key for key in keys_of(i.d)] if i.d is not None else []
            if i.op == "v_permlane32_swap":
                wr = rd
            for key in rd:
                if key in last_w:
                    preds[k].add(last_w[key])
            for key in wr:
                if key in last_w:
                    preds[k].add(last_w[key])
                for r in readers.get(key, ()):
                    if r != k:
                        preds[k].add(r)
            for key in rd:
                readers.setdefault(key, []).append(k)
            for key in wr:
                last_w[key] = k
                readers[key] = []
            if i.op.startswith("ds_"):
                if last_mem is not None and last_mem not in preds[k]:
                    chain[k] = last_mem          # issue order only (one slot), not a data dependence
                last_mem = k
        succs = [[] for _ in range(n)]
        for k in range(n):
            for p_ in preds[k]:
                succs[p_].append(k)
        npred = [len(p_) + (1 if chain[k] is not None else 0) for k, p_ in enumerate(preds)]
        chain_succ = [None] * n
        for k in range(n):
            if chain[k] is not None:
                chain_succ[chain[k]] = k
        ready_at = [0] * n            # earliest slot at which all operands are available
        done = [False] * n
        order = []
        lo = 0                         # lowest unscheduled original index
        slot = 0
        stalls = 0
        while len(order) < n:
            while lo < n and done[lo]:
                lo += 1
            cand = [k for k in range(lo, min(n, lo + window)) if not done[k] and npred[k] == 0]
            good = [k for k in cand if ready_at[k] <= slot]
            if good:
                k = good[0]
            else:
                k = min(cand, key=lambda c: (ready_at[c], c))
                stalls += ready_at[k] - slot
                slot = ready_at[k]
            done[k] = True
            order.append(ins[k])
            lat = self.LAT_LDS if ins[k].op == "ds_read_b64" else (self.LAT_VALU if ins[k].op.startswith("v_") else 1)
            for s_ in succs[k]:
                npred[s_] -= 1
                ready_at[s_] = max(ready_at[s_], slot + lat)
            if chain_succ[k] is not None:
                npred[chain_succ[k]] -= 1
                ready_at[chain_succ[k]] = max(ready_at[chain_succ[k]], slot + 1)
            slot += 1
        self.stalls = stalls
        self.ins = order

    # ---- register allocation over the scheduled order
    def allocate(self):
        xp, hh = self.in_layout()
        last = {}
        for k, i in enumerate(self.ins):
            for o in i.operands():
                if o[0] in ("v", "h"):
                    last[o[1]] = k
        free = sorted(p for p in range(self.POOL0, self.POOL0 + 2 * self.NPOOL, 2))
        inputs = {v.p for v in xp} | {v.p for g in hh for v in g}
        free = [p for p in free if p not in inputs]
        pmap = {}
        quar = deque()
        live = len(inputs)
        for k, i in enumerate(self.ins):
            while quar and quar[0][0] <= k:
                free.append(quar.popleft()[1])
            for o in i.operands():
                if o[0] in ("v", "h") and o[1] >= VBASE and o[1] not in pmap:
                    if not free:
                        if quar:
                            free.append(quar.popleft()[1])
                        else:
                            raise RuntimeError(f"register pool exhausted at instruction {k}")
                    late = [q_ for q_ in free if q_ not in inputs]          # pairs that are not a unit's operand registers first: those
                    p = min(late) if late else min(free)                    # stay free from their last use on, and the next unit's loads move in early
                    free.remove(p)
                    pmap[o[1]] = p
                    live += 1
                    self.maxlive = max(self.maxlive, live)
            for o in i.operands():
                if o[0] in ("v", "h") and last.get(o[1]) == k:
                    phys = pmap[o[1]] if o[1] >= VBASE else o[1]
                    if self.POOL0 <= phys < self.POOL0 + 2 * self.NPOOL and (phys, k) not in [(q[1], k) for q in quar]:
                        quar.append((k + 3, phys))
                        last[o[1]] = -1
                        live -= 1
        # rewrite to physical operands
        def ph(o):
            if o is None:
                return None
            if o[0] == "v" and o[1] >= VBASE:
                return ("v", pmap[o[1]])
            if o[0] == "h" and o[1] >= VBASE:
                return ("h", pmap[o[1]], o[2])
            return o
        for i in self.ins:
            i.d = ph(i.d)
            i.s = tuple(ph(o) for o in i.s)

    # ---- wait counts: before the first instruction that touches the destination of an outstanding LDS read
    def insert_waits(self):
        out = []
        q = []          # outstanding LDS operations, oldest first: set of register units or None (stores)
        for i in self.ins:
            regs = {key for o in i.operands() for key in keys_of(o)}
            need = -1
            for pos, dst in enumerate(q):
                if dst and dst & regs:
                    need = pos
            if need >= 0:
                cnt = min(len(q) - 1 - need, 14)      # four bits, and 15 means "do not wait"
                out.append(Ins("s_waitcnt_lgkm", imm=cnt))
                del q[: len(q) - cnt]
            out.append(i)
            if i.op == "ds_read_b64":
                q.append(set(keys_of(i.d)))
            elif i.op == "ds_write_b64":
                q.append(None)
        self.ins = out

    # ---- permlane hazard: a VALU write of a swap operand needs two wait states before the swap reads it
    def fix_permlane_hazards(self):
        out = []
        for ins in self.ins:
            if ins.op == "v_permlane32_swap":
                regs = {key for o in ins.s for key in keys_of(o)}
                need, dist = 0, 0
                for prev in reversed(out):
                    if dist >= 2:
                        break
                    wr = set()
                    if prev.op == "v_permlane32_swap":
                        wr = {key for o in prev.s for key in keys_of(o)}
                    elif prev.d is not None and prev.op.startswith("v_"):
                        wr = set(keys_of(prev.d))
                    if wr & regs:
                        need = max(need, 2 - dist)
                    dist += prev.imm + 1 if prev.op == "s_nop" else 1
                if need:
                    out.append(Ins("s_nop", imm=need - 1))
            out.append(ins)
        self.ins = out

    def merge_loads(self, body, loads, gap=4):
        """Spread the NEXT unit's loads through the tail of this unit's body: a load may be issued once its destination pair has been touched for the
        last time.  A burst of 96 VMEM instructions at the end of the unit kept the wave ~2000 clocks in the issue queue of the memory pipeline
        (ablation: 0.18 of 1.13 ms); between arithmetic instructions each load finds the queue empty."""
        last = {}
        for k, i in enumerate(body):
            for o in i.operands():
                if o[0] in ("v", "h"):
                    last[o[1]] = k
        pend = []
        for ld in loads:
            if ld.op != "buffer_load_dword":
                continue
            pend.append((last.get(ld.d[1], -1) + gap, ld))
        pend.sort(key=lambda t: t[0])
        out = []
        qi = 0
        since = 0
        n = len(body)
        for k, i in enumerate(body):
            out.append(i)
            since += 1
            remaining_loads = len(pend) - qi
            if remaining_loads == 0:
                continue
            stride = max(1, (n - k) // (remaining_loads + 1))
            if pend[qi][0] <= k and since >= min(stride, 6):
                out.append(pend[qi][1])
                qi += 1
                since = 0
        out.extend(ld for _, ld in pend[qi:])
        self.loads_in_body = qi
        return out

    def build_unit(self, window=64):
        """Emission -> schedule -> allocate -> waits -> hazards; returns the unit body (after the vmcnt(0) at its top)."""
        self.ins = []
        self.emit_unit()
        self.schedule(window)
        self.allocate()
        self.insert_waits()
        self.fix_permlane_hazards()
        body = self.ins
        loads = self.build_loads()
        body = self.merge_loads(body, loads)
        return [Ins("s_waitcnt_vm", imm=0)] + body

    def build_loads(self):
        self.ins = []
        self.emit_loads()
        return [i for i in self.ins]


# ------------------------------------------------------------------------------------------------------------------ emulator
class Emu:
    def __init__(self, g, sconst):
        self.v = np.zeros((256, 64), dtype=np.float32)
        self.vi = self.v.view(np.int32)
        self.s = sconst                         # dict: sgpr pair -> (lo, hi) float32
        self.lds = np.zeros(LDS_BYTES // 4, dtype=np.float32)
        self.glob = None
        self.gbase = 0

    def reg(self, o):
        """index of the 32-bit register an ('h', p, k) / ('r', n) operand names"""
        return o[1] + o[2] if o[0] == "h" else o[1]

    def src(self, s, mods, i, which):
        sel = mods["op_sel"][i] if which == "lo" else mods["op_sel_hi"][i]
        neg = mods["neg_lo"][i] if which == "lo" else mods["neg_hi"][i]
        if s[0] == "v":
            x = self.v[s[1] + sel]
        else:
            x = np.full(64, self.s[s[1]][sel], dtype=np.float32)
        return -x if neg else x

    def run(self, ins_list):
        for ins in ins_list:
            op = ins.op
            if op in ("comment", "s_waitcnt_lgkm", "s_waitcnt_vm", "s_nop"):
                continue
            if op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"):
                res = []
                for which in ("lo", "hi"):
                    xs = [self.src(s, ins.mods, i, which) for i, s in enumerate(ins.s)]
                    if op == "v_pk_add_f32":
                        r = xs[0] + xs[1]
                    elif op == "v_pk_mul_f32":
                        r = xs[0] * xs[1]
                    else:
                        r = (xs[0].astype(np.float64) * xs[1].astype(np.float64) + xs[2].astype(np.float64)).astype(np.float32)   # fused: one rounding
                    res.append(r.astype(np.float32))
                self.v[ins.d[1]] = res[0]
                self.v[ins.d[1] + 1] = res[1]
            elif op == "v_fma_f32":
                a, b, c = (self.v[self.reg(s)] for s in ins.s)
                self.v[self.reg(ins.d)] = (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
            elif op == "ds_read_b64":
                addr = self.vi[self.reg(ins.s[0])] + ins.imm
                assert np.all(addr % 8 == 0)
                self.v[ins.d[1]] = self.lds[addr // 4]
                self.v[ins.d[1] + 1] = self.lds[addr // 4 + 1]
            elif op == "ds_write_b64":
                addr = self.vi[self.reg(ins.s[0])] + ins.imm
                assert np.all(addr % 8 == 0) and len(set(addr.tolist())) == 64
                self.lds[addr // 4] = self.v[ins.s[1][1]]
                self.lds[addr // 4 + 1] = self.v[ins.s[1][1] + 1]
            elif op == "buffer_load_dword":
                addr = self.gbase + ins.imm + self.vi[self.reg(ins.s[0])]
                self.v[self.reg(ins.d)] = self.glob[addr // 4]
            elif op == "v_permlane32_swap":
                a, b = self.reg(ins.s[0]), self.reg(ins.s[1])
                ta = self.v[a].copy()
                self.v[a, 32:] = self.v[b, :32]
                self.v[b, :32] = ta[32:]
            else:
                raise RuntimeError("emulator: unknown op " + op)


def w64(m):
    a = -2.0 * math.pi * m / 64
    return np.float32(math.cos(a)), np.float32(math.sin(a))


def sconsts(g):
    h = np.float32(math.sqrt(0.5))
    sc = {g.S_HH: (h, h), g.S_PM: (np.float32(1.0), np.float32(-1.0))}
    for m, p in g.wexp.items():
        sc[p] = w64(m)
    return sc


def check(window=64):
    rng = np.random.default_rng(1776)
    g = Gen()
    unit = g.build_unit(window)
    loads = g.build_loads()
    body = unit
    nunits = 3
    sig = rng.standard_normal((nunits + 3) * N).astype(np.float32)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / (N - 1))).astype(np.float32)
    em = Emu(g, sconsts(g))
    lane = np.arange(64)
    em.vi[g.V_OFF] = lane * 4
    em.vi[g.V_WIN] = lane * 8
    wave = 3
    xb = WIN_BYTES + wave * XBUF_BYTES
    em.vi[g.V_XW] = xb + ((lane >> 5) * 32 + (lane & 31)) * XROW * 8
    em.vi[g.V_XR] = xb + ((lane >> 5) * 32) * XROW * 8 + (lane & 31) * 8
    wl = em.lds[: WIN_BYTES // 4].reshape(HALF, 2)
    wl[:, 0] = win[:HALF]
    wl[:, 1] = win[HALF:]
    roots = np.exp(-2j * np.pi * np.arange(N) / N)
    for j in range(1, 8):
        wa = roots[(8 * lane * j) % N]
        wb = roots[(lane * j) % N]
        em.v[g.TW0 + 2 * (j - 1)] = wa.real.astype(np.float32)
        em.v[g.TW0 + 2 * (j - 1) + 1] = wa.imag.astype(np.float32)
        em.v[g.TW0 + 14 + 2 * (j - 1)] = wb.real.astype(np.float32)
        em.v[g.TW0 + 14 + 2 * (j - 1) + 1] = wb.imag.astype(np.float32)
    em.glob = sig
    ref = np.zeros(N)
    em.v[g.POOL0:] = np.float32(np.nan)      # nothing may depend on what the pool held before
    em.gbase = 0
    em.run(loads)                            # the prologue's loads of the first unit
    for u in range(nunits):
        em.gbase = (u + 1) * N * 4           # the body carries the NEXT unit's loads
        em.run(body)
        a = sig[u * N: u * N + N].astype(np.float64)
        b = sig[u * N + HALF: u * N + HALF + N].astype(np.float64)
        Z = np.fft.fft(win.astype(np.float64) * (a + 1j * b))
        ref += np.abs(Z) ** 2
    got = np.zeros(N)
    for kt in range(64):
        got[lane + 64 * kt] = em.v[g.ACC0 + slot64(kt)]
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    worst = np.max(np.abs(got - ref) / ref.max())
    kinds = {}
    for i in body:
        kinds[i.op] = kinds.get(i.op, 0) + 1
    nins = sum(v for k, v in kinds.items() if k != "comment")
    nbad = verify_waits(loads + body + body)
    print(f"emulated {nunits} units: relerr {err:.3e}, worst bin / max {worst:.3e}; {nins} instructions per unit, peak live pairs {g.maxlive} of {g.NPOOL}, "
          f"scheduler stall slots {g.stalls}, {g.loads_in_body} of 96 loads inside the body, wait check: {nbad} problems")
    print("  ", {k: v for k, v in sorted(kinds.items()) if k != "comment"})
    return err < 2e-6 and nbad == 0


# ------------------------------------------------------------------------------------------------------------------ the kernel text
ABLATE = set()      # --ablate loads,lds,perm,acc : leave those instruction classes out of the emitted kernel (timing experiments; results are garbage)


def keep(ins):
    if "loads" in ABLATE and ins.op == "buffer_load_dword":
        return False
    if "lds" in ABLATE and ins.op in ("ds_read_b64", "ds_write_b64", "s_waitcnt_lgkm"):
        return False
    if "perm" in ABLATE and ins.op == "v_permlane32_swap":
        return False
    if "acc" in ABLATE and ins.op == "v_fma_f32":
        return False
    return True


def kernel_text():
    g = Gen()
    L = []
    A = L.append
    A('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
    A("\t.amdhsa_code_object_version 6")
    A("\t.text")
    A("\t.protected\tmdsp_welch_w64_asm")
    A("\t.globl\tmdsp_welch_w64_asm")
    A("\t.p2align\t8")
    A("\t.type\tmdsp_welch_w64_asm,@function")
    A("mdsp_welch_w64_asm:")
    A("; generated by tools/gen_welch_asm.py -- do not edit")
    # ---- prologue.  s[0:1] kernarg, s2 = workgroup x, s3 = workgroup y (channel), v0 = thread id
    A("\ts_load_dwordx8 s[4:11], s[0:1], 0x0          ; s, part, winpairs, tw")
    A("\ts_load_dwordx8 s[12:19], s[0:1], 0x20        ; lds_, units, run_len, nch")
    A("\ts_load_dword s20, s[0:1], 0x40               ; nflush")
    A("\tv_and_b32_e32 v1, 63, v0                     ; lane")
    A("\tv_lshrrev_b32_e32 v2, 6, v0                  ; wave")
    A("\ts_nop 1                                       ; gfx940+: a VALU write of a VGPR needs a wait state before v_readfirstlane reads it")
    A("\tv_readfirstlane_b32 s21, v2")
    A("\ts_nop 3                                       ; ... and the SGPR it writes a few before anything reads it")
    A("\ts_waitcnt lgkmcnt(0)")
    # window pairs: 16 KiB global -> LDS, 32 bytes per thread
    A("\tv_lshlrev_b32_e32 v3, 5, v0                  ; tid * 32")
    A("\tglobal_load_dwordx4 v[8:11], v3, s[8:9]")
    A("\tglobal_load_dwordx4 v[12:15], v3, s[8:9] offset:16")
    # per-lane twiddles: 28 floats at tw + lane * 112
    A("\tv_mul_u32_u24_e32 v4, 112, v1")
    for k in range(7):
        A(f"\tglobal_load_dwordx4 v[{Gen.TW0 + 4 * k}:{Gen.TW0 + 4 * k + 3}], v4, s[10:11] offset:{16 * k}")
    A("\ts_waitcnt vmcnt(7)")
    A("\tds_write_b128 v3, v[8:11]")
    A("\tds_write_b128 v3, v[12:15] offset:16")
    A("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    A("\ts_barrier")
    # addresses
    A(f"\tv_lshlrev_b32_e32 v{Gen.V_OFF}, 2, v1               ; lane * 4")
    A("\tv_lshrrev_b32_e32 v5, 5, v1                  ; lane >> 5")
    A("\tv_and_b32_e32 v6, 31, v1                     ; lane & 31")
    A(f"\ts_mul_i32 s22, s21, {XBUF_BYTES}")
    A(f"\ts_add_i32 s22, s22, {WIN_BYTES}                ; this wave's exchange buffer")
    A(f"\tv_lshl_add_u32 v7, v5, 5, v6                 ; (lane >> 5) * 32 + (lane & 31)")
    A(f"\tv_mul_u32_u24_e32 v7, {XROW * 8}, v7")
    A(f"\tv_add_u32_e32 v{Gen.V_XW}, s22, v7")
    A(f"\tv_mul_u32_u24_e32 v7, {32 * XROW * 8}, v5")
    A(f"\tv_lshl_add_u32 v7, v6, 3, v7")
    A(f"\tv_add_u32_e32 v{Gen.V_XR}, s22, v7")
    A(f"\tv_lshlrev_b32_e32 v{Gen.V_WIN}, 3, v1              ; lane * 8   (v1 was the lane: last use above)")
    # accumulators
    for s in range(64):
        A(f"\tv_mov_b32_e32 v{Gen.ACC0 + s}, 0")
    # constants
    for k in range(5):
        A(f"\ts_mov_b32 s{Gen.S_K4 + k}, {4096 * (k + 1)}")
    h = np.float32(math.sqrt(0.5))

    def fbits(x):
        return "0x%08x" % int(np.float32(x).view(np.uint32))

    A(f"\ts_mov_b32 s{Gen.S_HH}, {fbits(h)}")
    A(f"\ts_mov_b32 s{Gen.S_HH + 1}, {fbits(h)}")
    A(f"\ts_mov_b32 s{Gen.S_PM}, {fbits(1.0)}")
    A(f"\ts_mov_b32 s{Gen.S_PM + 1}, {fbits(-1.0)}")
    for m, p in g.wexp.items():
        c, s_ = w64(m)
        A(f"\ts_mov_b32 s{p}, {fbits(c)}                   ; W64^{m}")
        A(f"\ts_mov_b32 s{p + 1}, {fbits(s_)}")
    # slot = bx * 8 + wave; units [u0, uend)
    A("\ts_lshl_b32 s22, s2, 3")
    A("\ts_add_u32 s22, s22, s21                      ; slot")
    A("\ts_mul_i32 s23, s22, s16                      ; u0 = slot * run_len   (32-bit: < 2^31 units)")
    A("\ts_add_u32 s86, s23, s16")
    A("\ts_min_u32 s86, s86, s14                      ; uend")
    A("\ts_cmp_ge_u32 s23, s86")
    A("\ts_cbranch_scc1 .Lend")
    A("\ts_sub_u32 s87, s86, s23                      ; units of this wave")
    # sample descriptor: base = s + (ch * lds_ + u0 * 4096) * 4
    A("\ts_mul_i32 s88, s3, s12                       ; ch * lds_ (low)")
    A("\ts_mul_hi_u32 s89, s3, s12")
    A("\ts_mul_i32 s90, s3, s13")
    A("\ts_add_u32 s89, s89, s90")
    A("\ts_lshl_b64 s[88:89], s[88:89], 2")
    A("\ts_add_u32 s24, s4, s88")
    A("\ts_addc_u32 s25, s5, s89")
    A("\ts_mul_hi_u32 s89, s23, 0x4000")
    A("\ts_mul_i32 s88, s23, 0x4000                    ; u0 * 16384 bytes")
    A("\ts_add_u32 s24, s24, s88")
    A("\ts_addc_u32 s25, s25, s89")
    A("\ts_and_b32 s25, s25, 0xffff")
    A("\ts_mov_b32 s26, 0x6000                         ; three half-frames")
    A("\ts_mov_b32 s27, 0x00020000")
    # partial rows: part + ((slot * nch + ch) * nflush) * 16384 bytes
    A("\ts_mul_i32 s88, s22, s18")
    A("\ts_add_u32 s88, s88, s3")
    A("\ts_mul_i32 s88, s88, s20                      ; row index of this wave's first row")
    A("\ts_mul_hi_u32 s89, s88, 0x4000")
    A("\ts_mul_i32 s88, s88, 0x4000")
    A("\ts_add_u32 s28, s6, s88")
    A("\ts_addc_u32 s29, s7, s89")
    A("\ts_and_b32 s29, s29, 0xffff")
    A("\ts_mov_b32 s30, 0x4000")
    A("\ts_mov_b32 s31, 0x00020000")
    A(f"\ts_mov_b32 s91, {FLUSH}                        ; units until the next flush")
    # first unit's loads
    unit = g.build_unit()
    loads = g.build_loads()
    for ins in loads:
        if keep(ins):
            A("\t" + render(ins, {}))
    A(".Lunit:")
    A("\ts_waitcnt vmcnt(0)")
    A("\ts_add_u32 s24, s24, 0x4000                    ; the loads inside the body fetch the NEXT unit ...")
    A("\ts_addc_u32 s25, s25, 0")
    A("\ts_cmp_eq_u32 s87, 1")
    A("\ts_cselect_b32 s26, 0, 0x6000                  ; ... which does not exist behind this wave's last one: an empty descriptor returns zeros")
    for ins in unit[1:]:
        if keep(ins):
            A("\t" + render(ins, {}))
    nbody = sum(1 for i in unit if i.op != "comment")
    A("\ts_sub_u32 s87, s87, 1")
    A("\ts_cmp_eq_u32 s87, 0")
    A("\ts_cbranch_scc1 .Lflush")
    A("\ts_sub_u32 s91, s91, 1")
    A("\ts_cmp_lg_u32 s91, 0")
    A("\ts_cbranch_scc1 .Lunit")
    A(".Lflush:")
    A("; one row of Float32 sums: bin lane + 64 kt sits in accumulator slot64(kt)")
    for kt in range(64):
        k, imm = divmod(256 * kt, 4096)
        so = "0" if k == 0 else f"s{Gen.S_K4 + k - 1}"
        A(f"\tbuffer_store_dword v{Gen.ACC0 + slot64(kt)}, v{Gen.V_OFF}, s[28:31], {so} offen offset:{imm}")
    A("\ts_add_u32 s28, s28, 0x4000")
    A("\ts_addc_u32 s29, s29, 0")
    A("\ts_nop 4")
    for s in range(64):
        A(f"\tv_mov_b32_e32 v{Gen.ACC0 + s}, 0")
    A(f"\ts_mov_b32 s91, {FLUSH}")
    A("\ts_cmp_lg_u32 s87, 0")
    A("\ts_cbranch_scc1 .Lunit")
    A(".Lend:")
    A("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    A("\ts_endpgm")
    A("\t.section\t.rodata,\"a\",@progbits")
    A("\t.p2align\t6, 0x0")
    A("\t.amdhsa_kernel mdsp_welch_w64_asm")
    A(f"\t\t.amdhsa_group_segment_fixed_size {LDS_BYTES}")
    A("\t\t.amdhsa_private_segment_fixed_size 0")
    A("\t\t.amdhsa_kernarg_size 72")
    A("\t\t.amdhsa_user_sgpr_count 2")
    A("\t\t.amdhsa_user_sgpr_dispatch_ptr 0")
    A("\t\t.amdhsa_user_sgpr_queue_ptr 0")
    A("\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1")
    A("\t\t.amdhsa_user_sgpr_dispatch_id 0")
    A("\t\t.amdhsa_user_sgpr_kernarg_preload_length 0")
    A("\t\t.amdhsa_user_sgpr_kernarg_preload_offset 0")
    A("\t\t.amdhsa_user_sgpr_private_segment_size 0")
    A("\t\t.amdhsa_uses_dynamic_stack 0")
    A("\t\t.amdhsa_enable_private_segment 0")
    A("\t\t.amdhsa_system_sgpr_workgroup_id_x 1")
    A("\t\t.amdhsa_system_sgpr_workgroup_id_y 1")
    A("\t\t.amdhsa_system_sgpr_workgroup_id_z 0")
    A("\t\t.amdhsa_system_sgpr_workgroup_info 0")
    A("\t\t.amdhsa_system_vgpr_workitem_id 0")
    A("\t\t.amdhsa_next_free_vgpr 256")
    A("\t\t.amdhsa_next_free_sgpr 96")
    A("\t\t.amdhsa_accum_offset 256")
    A("\t\t.amdhsa_reserve_vcc 0")
    A("\t\t.amdhsa_float_round_mode_32 0")
    A("\t\t.amdhsa_float_round_mode_16_64 0")
    A("\t\t.amdhsa_float_denorm_mode_32 3")
    A("\t\t.amdhsa_float_denorm_mode_16_64 3")
    A("\t\t.amdhsa_dx10_clamp 1")
    A("\t\t.amdhsa_ieee_mode 1")
    A("\t\t.amdhsa_fp16_overflow 0")
    A("\t\t.amdhsa_tg_split 0")
    A("\t.end_amdhsa_kernel")
    A("\t.text")
    A(".Lfunc_end0:")
    A("\t.size\tmdsp_welch_w64_asm, .Lfunc_end0-mdsp_welch_w64_asm")
    A("\t.amdgpu_metadata")
    A("---")
    A("amdhsa.kernels:")
    A("  - .agpr_count:     0")
    A("    .args:")
    A("      - .offset:         0")
    A("        .size:           72")
    A("        .value_kind:     by_value")
    A(f"    .group_segment_fixed_size: {LDS_BYTES}")
    A("    .kernarg_segment_align: 8")
    A("    .kernarg_segment_size: 72")
    A("    .language:       OpenCL C")
    A("    .language_version:")
    A("      - 2")
    A("      - 0")
    A("    .max_flat_workgroup_size: 512")
    A("    .name:           mdsp_welch_w64_asm")
    A("    .private_segment_fixed_size: 0")
    A("    .sgpr_count:     102")
    A("    .sgpr_spill_count: 0")
    A("    .symbol:         mdsp_welch_w64_asm.kd")
    A("    .uniform_work_group_size: 1")
    A("    .uses_dynamic_stack: false")
    A("    .vgpr_count:     256")
    A("    .vgpr_spill_count: 0")
    A("    .wavefront_size: 64")
    A("amdhsa.target:   amdgcn-amd-amdhsa--gfx950")
    A("amdhsa.version:")
    A("  - 1")
    A("  - 2")
    A("...")
    A("")
    A("\t.end_amdgpu_metadata")
    return "\n".join(L) + "\n", nbody


def verify_waits(ins_list):
    """Independent check of the wait counts: replay the instruction list keeping the in-order queues of outstanding LDS and VMEM operations; every
    register an instruction reads (or overwrites) must not be the destination of an operation still in its queue."""
    lds, vm = [], []          # destination register units of outstanding operations, oldest first (None for stores)
    bad = 0
    for n, ins in enumerate(ins_list):
        if ins.op == "comment":
            continue
        if ins.op == "s_waitcnt_lgkm":
            while len(lds) > ins.imm:
                lds.pop(0)
            continue
        if ins.op == "s_waitcnt_vm":
            while len(vm) > ins.imm:
                vm.pop(0)
            continue
        if ins.op == "s_nop":
            continue
        regs = {key for o in ins.operands() for key in keys_of(o)}
        for q, name in ((lds, "LDS"), (vm, "VMEM")):
            for dst in q:
                if dst and dst & regs:
                    print(f"  wait missing: instruction {n} ({render(ins, {})}) touches the destination of an outstanding {name} operation")
                    bad += 1
        if ins.op == "ds_read_b64":
            lds.append(set(keys_of(ins.d)))
        elif ins.op == "ds_write_b64":
            lds.append(None)
        elif ins.op == "buffer_load_dword":
            vm.append(set(keys_of(ins.d)))
        elif ins.op == "buffer_store_dword":
            vm.append(None)
    return bad



if __name__ == "__main__":
    import os
    if "--check" in sys.argv:
        sys.exit(0 if check() else 1)
    if "--window" in sys.argv:
        w = int(sys.argv[sys.argv.index("--window") + 1])
        sys.exit(0 if check(w) else 1)
    if "--ablate" in sys.argv:
        ABLATE.update(sys.argv[sys.argv.index("--ablate") + 1].split(","))
    text, nbody = kernel_text()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp.jl_amd", "csrc", "welch_w64_asm.s")
    open(out, "w").write(text)
    print(f"wrote {out}: {text.count(chr(10))} lines, {nbody} instructions per unit")


