#!/usr/bin/env python3
"""Generator (and CPU emulator) of the hand-allocated Welch kernel `mdsp_welch_w64_asm` (round 4).

Why a generator: the one-wavefront-per-transform factorisation (4096 = 64 x 64, ONE exchange, no barrier; csrc/fft_w64.h has the algebra and
tests/cpu_harness/fft_emul.cpp the host proof) only pays with TWO waves per SIMD -- a single wave issues a packed instruction every 5.7 clocks,
two every 4.75 (profiles/r02t_valu_rate.txt; variant 40 measured 1.65 ms against 1.36) -- and two waves per SIMD means <= 256 registers per
wave for 64 complex points (128) + 64 power accumulators + 28 twiddle values + the butterflies' temporaries.  hipcc needs 256 + 129 spilled
(variant 41: 2.17 ms).  Here every register is assigned by this script: values are allocated from a pool of 80 register pairs and freed at
their last use, which the script knows because it emits the dataflow itself.

What it writes:  dsp.jl_amd/csrc/welch_w64_asm.s  (assembled by build.py with clang -x assembler, linked by ld.lld, loaded with
hipModuleLoadData by csrc/asmkernels.hip).  `--check` runs the emitted instruction list through a lane-level emulator of the ~15 opcodes it
uses (numpy, one wavefront) for two consecutive units and compares the accumulators with numpy's FFT of the windowed frame pairs.

Kernel contract (csrc/asmkernels.hip fills the arguments):
    struct W64AsmArgs { const float* s; float* part; const float* winpairs; const float* tw; int64 lds_, units, run_len, nch; int nflush, pad; }
    grid (G, nch), 512 threads = 8 independent waves; wave w of workgroup b is slot 8 b + w and owns units [slot run_len, (slot+1) run_len) of
    its channel (a unit = two frames = 4096 new samples; every unit handed to this kernel has BOTH frames -- the odd last frame of a channel
    goes through welch_half3_kernel); every 128 units (and at the end) the 64 Float32 sums per lane are stored as one row of
    part[((slot nch + ch) nflush + f) 4096 + bin] -- rows that are never written stay at the zeros the launcher put there.
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
class Ins:
    __slots__ = ("op", "d", "s", "mods", "text", "imm", "extra")

    def __init__(self, op, d=None, s=(), mods=None, text="", imm=0, extra=None):
        self.op, self.d, self.s, self.mods, self.text, self.imm, self.extra = op, d, tuple(s), mods or {}, text, imm, extra


def vp(p):           # a 64-bit VGPR pair operand
    return f"v[{p}:{p + 1}]"


def sp(p):
    return f"s[{p}:{p + 1}]"


def modtext(mods, nsrc):
    out = []
    for key in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"):
        if key in mods:
            v = list(mods[key])[:nsrc]
            dflt = [1] * nsrc if key == "op_sel_hi" else [0] * nsrc
            if v != dflt:
                out.append(f"{key}:[{','.join(str(x) for x in v)}]")
    return (" " + " ".join(out)) if out else ""


class Val:
    """A complex (pair) value living in VGPR pair `p`; `.lo` / `.hi` are its 32-bit registers."""
    __slots__ = ("p", "pend")

    def __init__(self, p):
        self.p = p
        self.pend = None       # sequence number of the LDS read that fills it (None once waited for)

    @property
    def lo(self):
        return self.p

    @property
    def hi(self):
        return self.p + 1


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
        self.free = deque(range(self.POOL0, self.POOL0 + 2 * self.NPOOL, 2))
        self.quar = deque()            # (release_at_instruction_index, pair)
        self.lds_seq = 0               # LDS operations issued so far
        self.pending = {}              # pair -> lds sequence number of the read that fills it
        self.maxlive = 0
        self.wexp = {m: self.S_W + 2 * i for i, m in enumerate(self.W_EXPS)}

    # ---- emission
    def emit(self, op, d=None, s=(), mods=None, text="", imm=0, extra=None):
        self.ins.append(Ins(op, d, s, mods, text, imm, extra))
        while self.quar and self.quar[0][0] <= len(self.ins):
            self.free.append(self.quar.popleft()[1])

    def comment(self, t):
        self.ins.append(Ins("comment", text="; " + t))

    def alloc(self):
        if not self.free:
            # let quarantined registers out early if the pool ran dry (they are at least one instruction old)
            if self.quar:
                self.free.append(self.quar.popleft()[1])
            else:
                raise RuntimeError("register pool exhausted")
        p = min(self.free)
        self.free.remove(p)
        live = self.NPOOL - len(self.free) - len(self.quar)
        self.maxlive = max(self.maxlive, live)
        return Val(p)

    def kill(self, *vals):
        for v in vals:
            assert v.p not in self.pending, "freeing a value whose load was never waited for"
            self.quar.append((len(self.ins) + 4, v.p))

    def use(self, *vals):
        """Wait for the LDS reads that fill these values (in-order return: lgkmcnt = operations issued after the youngest one needed)."""
        need = [self.pending[v.p] for v in vals if isinstance(v, Val) and v.p in self.pending]
        if not need:
            return
        youngest = max(need)
        cnt = min(self.lds_seq - youngest, 14)   # four bits, and 15 means "do not wait": 14 is the largest wait the encoding can express (conservative beyond)
        self.emit("s_waitcnt_lgkm", imm=cnt, text=f"s_waitcnt lgkmcnt({cnt})")
        for p, q in list(self.pending.items()):
            if q <= youngest:
                del self.pending[p]

    # ---- packed arithmetic: every source is ('v', pair) | ('s', pair)
    def _src(self, x):
        if isinstance(x, Val):
            return ("v", x.p)
        return x

    def _srctext(self, s):
        return vp(s[1]) if s[0] == "v" else sp(s[1])

    def pk(self, op, srcs, mods=None, dst=None):
        self.use(*[x for x in srcs if isinstance(x, Val)])
        d = dst or self.alloc()
        ss = [self._src(x) for x in srcs]
        n = len(ss)
        mods = dict(mods or {})
        for key, dflt in (("op_sel", 0), ("op_sel_hi", 1), ("neg_lo", 0), ("neg_hi", 0)):
            v = list(mods.get(key, [dflt] * n))
            v += [dflt] * (n - len(v))
            mods[key] = v
        text = f"{op} {vp(d.p)}, " + ", ".join(self._srctext(s) for s in ss) + modtext(mods, n)
        self.emit(op, ("v", d.p), ss, mods, text)
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
        """a W64^m (forward root), m = n1 k1 < 64.  Consumes `a` (frees it) unless m == 0."""
        if m == 0:
            return a
        if m == 16:              # -i a = (a.y, -a.x)
            d = self.pk("v_pk_mul_f32", [a, ("s", self.S_PM)], {"op_sel": [1, 0], "op_sel_hi": [0, 1]})
        elif m == 8:             # h (a - i a)
            t = self.sub_ib(a, a)
            d = self.pk("v_pk_mul_f32", [t, ("s", self.S_HH)], dst=t)
        elif m == 24:            # -h (a + i a)
            t = self.add_ib(a, a)
            d = self.pk("v_pk_mul_f32", [t, ("s", self.S_HH)], {"neg_lo": [1, 0], "neg_hi": [1, 0]}, dst=t)
        else:
            d = self.cmul(a, ("s", self.wexp[m]))
        self.kill(a)
        return d

    # ---- butterflies (forward), consuming their inputs
    def bfly4(self, a0, a1, a2, a3):
        t0 = self.add(a0, a2)
        t2 = self.add(a1, a3)
        t1 = self.sub(a0, a2)
        d = self.sub(a1, a3)
        self.kill(a0, a1, a2, a3)
        x0 = self.add(t0, t2)
        x2 = self.sub(t0, t2)
        x1 = self.sub_ib(t1, d)
        x3 = self.add_ib(t1, d)
        self.kill(t0, t2, t1, d)
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
        self.kill(e[0], o[0], o[1], o[3], e[2], o[2])
        out[1] = self.axpy(u1, e[1])
        out[5] = self.axmy(u1, e[1])
        out[3] = self.axmy(u3, e[3])
        out[7] = self.axpy(u3, e[3])
        self.kill(u1, u3, e[1], e[3])
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
        self.kill(*S)
        e1 = self.sub_ib(D[0], D[2])
        e3 = self.add_ib(D[0], D[2])
        o1 = self.sub_ib(D[1], D[3])
        o3 = self.add_ib(D[1], D[3])
        self.kill(*D)
        return self.bfly8_tail([e0, e1, e2, e3], [o0, o1, o2, o3])

    def bfly64_tail(self, v):
        for k1 in range(8):
            u = [v[j + 8 * k1] for j in range(8)]
            u = [u[0]] + [self.mul_w64(u[j], (j * k1) & 63) for j in range(1, 8)]
            out = self.bfly8(u)
            for k2 in range(8):
                v[k2 + 8 * k1] = out[k2]

    # ---- memory
    def ds_read(self, addr_vgpr, offset, dst=None):
        d = dst or self.alloc()
        self.emit("ds_read_b64", ("v", d.p), [("v32", addr_vgpr)], imm=offset, text=f"ds_read_b64 {vp(d.p)}, v{addr_vgpr} offset:{offset}")
        self.lds_seq += 1
        self.pending[d.p] = self.lds_seq
        return d

    def ds_write(self, addr_vgpr, offset, src):
        self.use(src)
        self.emit("ds_write_b64", None, [("v32", addr_vgpr), ("v", src.p)], imm=offset, text=f"ds_write_b64 v{addr_vgpr}, {vp(src.p)} offset:{offset}")
        self.lds_seq += 1

    def buffer_load(self, dst32, off_bytes):
        k, imm = divmod(off_bytes, 4096)
        so = "0" if k == 0 else f"s{self.S_K4 + k - 1}"
        self.emit("buffer_load_dword", ("v32", dst32), [("v32", self.V_OFF)], imm=off_bytes,
                  text=f"buffer_load_dword v{dst32}, v{self.V_OFF}, s[{self.S_RS}:{self.S_RS + 3}], {so} offen offset:{imm}")

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
        inpairs = {v.p for v in xp} | {v.p for g in hh for v in g}
        self.free = deque(p for p in range(self.POOL0, self.POOL0 + 2 * self.NPOOL, 2) if p not in inpairs)
        self.quar.clear()
        self.emit("s_waitcnt_vm", imm=0, text="s_waitcnt vmcnt(0)")
        v = [None] * 64
        self.comment("pass A, first layer (window folded in) + first radix-8 layer")

        def fetch_win(n1):
            return [self.ds_read(self.V_WIN, 512 * (n1 + 8 * j)) for j in range(4)]

        wp = fetch_win(0)
        for n1 in range(8):
            wnext = fetch_win(n1 + 1) if n1 < 7 else None
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
                self.kill(xp[e], wp[j])
                if h:
                    self.kill(hh[n1][j >> 1])
            o = self.bfly8_sd(S, D)
            for k1 in range(8):
                v[n1 + 8 * k1] = o[k1]
            wp = wnext
        self.comment("pass A, second radix-8 layer (W64 roots from SGPR pairs)")
        self.bfly64_tail(v)           # v[slot64(ke)] = Y_t[ke]
        self.comment("64 x 64 transposition: half exchange in registers, then two rounds of 32 x 32 through LDS")
        m = [v[slot64(r)] for r in range(64)]
        for r in range(32):
            for half in ("lo", "hi"):
                a, b = getattr(m[r], half), getattr(m[r + 32], half)
                self.emit("v_permlane32_swap", None, [("v32", a), ("v32", b)], text=f"v_permlane32_swap_b32_e32 v{a}, v{b}")
        nv = [None] * 64
        for rnd in range(2):
            for r in range(32):
                self.ds_write(self.V_XW, 8 * r, m[32 * rnd + r])
                self.kill(m[32 * rnd + r])
            # read order: the operands of pass B's first groups first
            for T in sorted(range(32), key=lambda T: (T & 7, T >> 3)):
                nv[32 * rnd + T] = self.ds_read(self.V_XR, 8 * XROW * T)
        v = nv
        self.comment("pass B: two-level twiddles W^{8 lane t2} in front of, W^{lane t1} behind the first radix-8 layer")
        for t1 in range(8):
            q = [v[t1]] + [None] * 7
            for t2 in range(1, 8):
                q[t2] = self.cmul(v[t1 + 8 * t2], Val(self.TW0 + 2 * (t2 - 1)))
                self.kill(v[t1 + 8 * t2])
            out = self.bfly8(q)
            for k1 in range(8):
                if t1 == 0:
                    v[t1 + 8 * k1] = out[k1]
                else:
                    v[t1 + 8 * k1] = self.cmul(out[k1], Val(self.TW0 + 14 + 2 * (t1 - 1)))
                    self.kill(out[k1])
        self.bfly64_tail(v)           # v[slot64(kt)] = X[lane + 64 kt]
        self.comment("power: acc[s] += re^2 + im^2")
        for s in range(64):
            a = self.ACC0 + s
            self.use(v[s])
            self.emit("v_fma_f32", ("v32", a), [("v32", v[s].lo), ("v32", v[s].lo), ("v32", a)], text=f"v_fma_f32 v{a}, v{v[s].lo}, v{v[s].lo}, v{a}")
            self.emit("v_fma_f32", ("v32", a), [("v32", v[s].hi), ("v32", v[s].hi), ("v32", a)], text=f"v_fma_f32 v{a}, v{v[s].hi}, v{v[s].hi}, v{a}")
        for s in range(64):
            self.kill(v[s])
        assert not self.pending

    # ---- permlane hazard: a VALU write of a swap operand needs two wait states before the swap reads it
    def fix_permlane_hazards(self):
        out = []
        for i, ins in enumerate(self.ins):
            if ins.op == "v_permlane32_swap":
                regs = {s[1] for s in ins.s}
                need = 0
                dist = 0
                for prev in reversed(out):
                    if prev.op == "comment":
                        continue
                    if dist >= 2:
                        break
                    wr = set()
                    if prev.d is not None:
                        wr = {prev.d[1], prev.d[1] + 1} if prev.d[0] == "v" else {prev.d[1]}
                    if prev.op == "v_permlane32_swap":
                        wr = {s[1] for s in prev.s}
                    if wr & regs:
                        need = max(need, 2 - dist)
                    dist += prev.imm + 1 if prev.op == "s_nop" else 1
                if need:
                    out.append(Ins("s_nop", imm=need - 1, text=f"s_nop {need - 1}"))
            out.append(ins)
        self.ins = out


# ------------------------------------------------------------------------------------------------------------------ emulator
class Emu:
    def __init__(self, g, sconst):
        self.v = np.zeros((256, 64), dtype=np.float32)
        self.vi = self.v.view(np.int32)
        self.s = sconst                         # dict: sgpr pair -> (lo, hi) float32
        self.lds = np.zeros(LDS_BYTES // 4, dtype=np.float32)
        self.glob = None
        self.gbase = 0

    def src(self, s, half, mods, i, which):
        sel = mods["op_sel"][i] if which == "lo" else mods["op_sel_hi"][i]
        neg = mods["neg_lo"][i] if which == "lo" else mods["neg_hi"][i]
        if s[0] == "v":
            x = self.v[s[1] + sel]
        else:
            x = np.full(64, self.s[s[1]][sel], dtype=np.float32)
        return -x if neg else x

    def run(self, ins_list):
        lane = np.arange(64)
        for ins in ins_list:
            op = ins.op
            if op in ("comment", "s_waitcnt_lgkm", "s_waitcnt_vm", "s_nop"):
                continue
            if op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"):
                res = []
                for which in ("lo", "hi"):
                    xs = [self.src(s, None, ins.mods, i, which) for i, s in enumerate(ins.s)]
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
                a, b, c = (self.v[s[1]] for s in ins.s)
                self.v[ins.d[1]] = (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
            elif op == "ds_read_b64":
                addr = self.vi[ins.s[0][1]] + ins.imm
                assert np.all(addr % 8 == 0)
                self.v[ins.d[1]] = self.lds[addr // 4]
                self.v[ins.d[1] + 1] = self.lds[addr // 4 + 1]
            elif op == "ds_write_b64":
                addr = self.vi[ins.s[0][1]] + ins.imm
                assert np.all(addr % 8 == 0) and len(set(addr.tolist())) == 64
                self.lds[addr // 4] = self.v[ins.s[1][1]]
                self.lds[addr // 4 + 1] = self.v[ins.s[1][1] + 1]
            elif op == "buffer_load_dword":
                addr = self.gbase + ins.imm + self.vi[ins.s[0][1]]
                self.v[ins.d[1]] = self.glob[addr // 4]
            elif op == "v_permlane32_swap":
                a, b = ins.s[0][1], ins.s[1][1]
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


def check():
    rng = np.random.default_rng(1776)
    g = Gen()
    g.emit_loads()
    g.emit_unit()
    g.fix_permlane_hazards()
    body = list(g.ins)
    nunits = 3
    sig = rng.standard_normal((nunits + 1) * N).astype(np.float32)
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
    for u in range(nunits):
        em.gbase = u * N * 4
        em.run(body)          # loads of unit u (the kernel issues them at the end of the previous unit), then the unit
        a = sig[u * N: u * N + N].astype(np.float64)
        b = sig[u * N + HALF: u * N + HALF + N].astype(np.float64)
        Z = np.fft.fft(win.astype(np.float64) * (a + 1j * b))
        ref += np.abs(Z) ** 2
    got = np.zeros(N)
    for kt in range(64):
        got[lane + 64 * kt] = em.v[g.ACC0 + slot64(kt)]
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    worst = np.max(np.abs(got - ref) / ref.max())
    nins = sum(1 for i in body if i.op != "comment")
    kinds = {}
    for i in body:
        kinds[i.op] = kinds.get(i.op, 0) + 1
    print(f"emulated {nunits} units: relerr {err:.3e}, worst bin / max {worst:.3e}; {nins} instructions per unit, peak live pairs {g.maxlive} of {g.NPOOL}")
    print("  ", {k: v for k, v in sorted(kinds.items()) if k != "comment"})
    return err < 2e-6


# ------------------------------------------------------------------------------------------------------------------ the kernel text
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
    g.emit_loads()
    for ins in g.ins:
        A("\t" + ins.text)
    g.ins = []
    A(".Lunit:")
    g.emit_unit()
    g.fix_permlane_hazards()
    for ins in g.ins:
        A("\t" + ins.text)
    nbody = sum(1 for i in g.ins if i.op != "comment")
    g.ins = []
    # advance to the next unit; its loads go into the registers the spectrum has just left
    A("\ts_add_u32 s24, s24, 0x4000")
    A("\ts_addc_u32 s25, s25, 0")
    A("\ts_sub_u32 s87, s87, 1")
    A("\ts_cmp_eq_u32 s87, 0")
    A("\ts_cbranch_scc1 .Lflush")
    g.emit_loads()
    for ins in g.ins:
        A("\t" + ins.text)
    g.ins = []
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


if __name__ == "__main__":
    import os
    if "--check" in sys.argv:
        sys.exit(0 if check() else 1)
    if "--verify" in sys.argv:
        sys.exit(0)
    text, nbody = kernel_text()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp.jl_amd", "csrc", "welch_w64_asm.s")
    open(out, "w").write(text)
    print(f"wrote {out}: {text.count(chr(10))} lines, {nbody} instructions per unit")


def verify_waits(ins_list):
    """Independent check of the wait counts: replay the instruction list keeping the in-order queues of outstanding LDS and VMEM operations; every
    register an instruction reads (or overwrites) must not be the destination of an operation still in its queue."""
    lds, vm = [], []          # destination register sets of outstanding operations, oldest first (None for stores)
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
        regs = set()
        for s in ins.s:
            if s[0] == "v":
                regs |= {s[1], s[1] + 1}
            elif s[0] == "v32":
                regs.add(s[1])
        if ins.d is not None:
            regs |= {ins.d[1], ins.d[1] + 1} if ins.d[0] == "v" else {ins.d[1]}
        for q, name in ((lds, "LDS"), (vm, "VMEM")):
            for dst in q:
                if dst and dst & regs:
                    print(f"  wait missing: instruction {n} ({ins.text}) touches {sorted(dst & regs)} of an outstanding {name} operation")
                    bad += 1
        if ins.op == "ds_read_b64":
            lds.append({ins.d[1], ins.d[1] + 1})
        elif ins.op == "ds_write_b64":
            lds.append(None)
        elif ins.op == "buffer_load_dword":
            vm.append({ins.d[1]})
    return bad


if __name__ == "__main__" and "--verify" in sys.argv:
    g = Gen()
    g.emit_loads()
    g.emit_unit()
    g.fix_permlane_hazards()
    one = list(g.ins)
    print("wait check over two consecutive units:", verify_waits(one + one), "problems")
