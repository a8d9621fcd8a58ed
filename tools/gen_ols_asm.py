#!/usr/bin/env python3
"""Generator (and CPU emulator) of the hand-allocated overlap-save kernel `mdsp_ols_w64_asm` (round 4) -- the headline filt / conv shape:
256 real Float32 taps, nfft = 2048, L = 1793 (optimalfftfiltlength(256, .), dspbase.jl:268-291; block loop Filters/filt.jl:504-518).

Same construction as tools/gen_welch_asm.py (whose instruction builder, list scheduler, register allocator, wait-count pass and emulator it
reuses): ONE wavefront owns a unit, here FOUR real blocks = two complex 2048-point transforms z = a + i b (real taps keep the two blocks of a
transform separable), every register assigned by the script, two waves per SIMD, no barrier in the loop.

    forward   2048 = 32 x 64:  lane t holds z_tau[t + 64 e], e < 32, of both transforms tau (64 register pairs)
              pass A   radix-32 over e per transform (in registers)            Y_tau,t[ke]
              exchange the 64 x 64 transposition of gen_welch_asm.py: lane L = ke + 32 tau receives Y_tau,t[ke] for t = 0..63
              twiddle  W2048^{t ke} as W^{8 ke t2} W^{ke t1} (two per-lane tables of seven roots), folded around the first radix-8 layer of
              pass B   64-point transform over t (in registers)                X_tau[ke + 32 kt] in lane (ke, tau), register kt
    multiply  by H[ke + 32 kt] (1/nfft folded in; a 16 KiB table in LDS), writing the product with its halves SWAPPED: with
              swap(u) = Im u + i Re u,  IFFT(P) = swap(FFT(swap(P)))  -- the inverse transform is the FORWARD transform of the swapped product, so it
              runs the same butterflies and the same root constants with the passes in reverse order:
    inverse   64-point transform over kt -> t, twiddle W^{t ke}, the transposition back, radix-32 over ke -> e: lane t ends with
              swap(z_tau)[t + 64 e] -- block a's outputs in the upper halves, block b's in the lower -- and stores 256 contiguous bytes per
              instruction (the nb - 1 = 255 aliased outputs in front of a block are never stored: e < 3 issue nothing, e = 3 stores lane 63 only).
Units handed to this kernel are INTERIOR (all four windows inside x, all outputs inside y, first block >= 4); the host runs the few edge blocks of a
column through ols_fused_kernel (csrc/ols.hip, ols_exec_core).

    struct OlsAsmArgs { const float* x; float* y; const float* H; const float* tw; int64 ldx, ldy, g_first, nunits, run_len; }
    grid (G, ncols), 512 threads = 8 independent waves; wave w of workgroup b is slot 8 b + w and owns units [slot run_len, (slot + 1) run_len) of the
    launch's nunits; unit u covers blocks g_first + 4 u .. + 3 of its column.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_welch_asm as W
from gen_welch_asm import Ins, Val, slot64, keys_of, VBASE, XROW, XBUF_BYTES

NFFT = 2048
NB = 256
L = NFFT - NB + 1            # 1793
LEAD = NB - 1                # 255 aliased outputs in front of a block
H_BYTES = NFFT * 8
LDS_BYTES = H_BYTES + 8 * XBUF_BYTES
UNIT_BYTES = 4 * L * 4       # 28688: bytes a unit advances in x and in y


class OGen(W.Gen):
    V_OFF, V_OFF3, V_XW, V_XR, V_H = 0, 1, 2, 3, 4
    TW0 = 6
    POOL0 = 34
    NPOOL = 111
    S_RX = 24      # s[24:27] x descriptor (based LEAD samples in front of the unit's first block)
    S_RY = 28      # s[28:31] y descriptor (based LEAD samples in front of the unit's first output)
    S_K4 = 32      # s32..s38 = 4096 k, k = 1..7
    S_HH = 40
    S_PM = 42
    S_W = 44

    def __init__(self):
        super().__init__()
        self.wexp = {m: self.S_W + 2 * i for i, m in enumerate(self.W_EXPS)}
        self.loads_in_body = 0

    # ---- extra arithmetic
    def cmul_swap(self, a, h):
        """swap(a h) = (Im(a h), Re(a h)): the product with its halves exchanged (see the header: the inverse transform runs forward on it)"""
        t = self.pk("v_pk_mul_f32", [a, h], {"op_sel": [1, 0], "op_sel_hi": [1, 1]})                                  # (a.y h.x, a.y h.y)
        return self.pk("v_pk_fma_f32", [a, h, t], {"op_sel": [0, 1, 0], "op_sel_hi": [0, 0, 1], "neg_hi": [0, 0, 1]}, dst=t)   # (a.x h.y + t.x, a.x h.x - t.y)

    def bfly32(self, v):
        """forward 32-point transform, natural order in -> natural order out (list of 32 values): n = n1 + 4 n2, k = k1 + 8 k2"""
        A = [[None] * 8 for _ in range(4)]
        for n1 in range(4):
            A[n1] = self.bfly8([v[n1 + 4 * n2] for n2 in range(8)])          # A[n1][k1]
        out = [None] * 32
        for k1 in range(8):
            u = [A[0][k1]] + [self.mul_w64(A[n1][k1], (2 * n1 * k1) & 63) for n1 in range(1, 4)]      # W32^{n1 k1} = W64^{2 n1 k1}
            x = self.bfly4(u[0], u[1], u[2], u[3])
            for k2 in range(4):
                out[k1 + 8 * k2] = x[k2]
        return out

    def bfly64(self, vnat):
        v = list(vnat)
        for n1 in range(8):
            u = self.bfly8([v[n1 + 8 * n2] for n2 in range(8)])
            for k1 in range(8):
                v[n1 + 8 * k1] = u[k1]
        self.bfly64_tail(v)
        return [v[slot64(k)] for k in range(64)]

    def transpose(self, m):
        """the 64 x 64 transposition on logical registers m[0..63] (consumes them), returns the new 64"""
        for r in range(32):
            for half in ("lo", "hi"):
                self.emit("v_permlane32_swap", None, [getattr(m[r], half), getattr(m[r + 32], half)])
        nv = [None] * 64
        order = sorted(range(32), key=lambda T: (T & 7, T >> 3))
        for rnd in range(2):
            for r in range(32):
                self.ds_write(self.V_XW, 8 * r, m[32 * rnd + r])
            for T in order:
                nv[32 * rnd + T] = self.ds_read(self.V_XR, 8 * XROW * T)
        return nv

    # ---- memory
    # Which pair holds which operand decides how long a unit waits for its predecessor's stores: the allocator reuses the LOWEST free operand pair when the
    # other registers run out, so low pairs stay busy until late in the unit, their loads for the next unit are issued last (among the stores), and
    # whoever reads them first waits for those stores.  Hence: the operands pass A consumes LAST live in the lowest pairs.
    @staticmethod
    def rank(tau, e):
        return 32 * tau + 8 * (e % 4) + e // 4          # consumption order of pass A (bfly32: n1 = e mod 4 outer, n2 = e div 4 inner)

    def inputs(self):
        return [[Val(self.POOL0 + 2 * (63 - self.rank(tau, e))) for e in range(32)] for tau in range(2)]

    def in_layout(self):
        return [v for row in self.inputs() for v in row], []

    @staticmethod
    def block_off(j, e):
        return j * L * 4 + 256 * e

    def emit_loads(self):
        D = self.inputs()
        self.comment("a unit's four block windows (2048 samples each, consecutive windows L = 1793 apart): 128 x 256-byte loads into the pairs (a, b) of pass A")
        for tau in range(2):
            for e in range(32):
                self.emit("buffer_load_dword", D[tau][e].lo, [("r", self.V_OFF)], imm=self.block_off(2 * tau, e))
                self.emit("buffer_load_dword", D[tau][e].hi, [("r", self.V_OFF)], imm=self.block_off(2 * tau + 1, e))

    def emit_unit(self):
        D = self.inputs()
        m = [None] * 64
        self.comment("pass A: radix-32 over e, both transforms")
        for tau in range(2):
            Y = self.bfly32(D[tau])
            for ke in range(32):
                m[ke + 32 * tau] = Y[ke]
        self.comment("transposition: lane (ke, tau) receives Y_t[ke], t = 0..63")
        v = self.transpose(m)
        self.comment("twiddles W2048^{t ke} = W^{8 ke t2} W^{ke t1} around the first radix-8 layer of the 64-point transform over t")
        tw = {}
        for t2 in range(1, 8):
            for t1 in range(8):
                tw[(t1, t2)] = self.cmul(v[t1 + 8 * t2], Val(self.TW0 + 2 * (t2 - 1)))
        for t1 in range(8):
            out = self.bfly8([v[t1]] + [tw[(t1, t2)] for t2 in range(1, 8)])
            for k1 in range(8):
                v[t1 + 8 * k1] = out[k1] if t1 == 0 else self.cmul(out[k1], Val(self.TW0 + 14 + 2 * (t1 - 1)))
        self.bfly64_tail(v)                     # v[slot64(kt)] = X[ke + 32 kt]
        self.comment("spectrum product, halves swapped; then the forward 64-point transform over kt (= the inverse transform of the product)")
        P = []
        for kt in range(64):
            h = self.ds_read(self.V_H, 8 * 32 * kt)
            P.append(self.cmul_swap(v[slot64(kt)], h))
        F1 = self.bfly64(P)                     # natural order over t
        self.comment("twiddles W^{t ke} again, the transposition back, radix-32 over ke")
        for t in range(64):
            t1, t2 = t & 7, t >> 3
            if t2:
                F1[t] = self.cmul(F1[t], Val(self.TW0 + 2 * (t2 - 1)))
            if t1:
                F1[t] = self.cmul(F1[t], Val(self.TW0 + 14 + 2 * (t1 - 1)))
        r = self.transpose(F1)                  # lane t: r[ke + 32 tau]
        self.comment("stores: block a of a transform sits in the upper halves, block b in the lower; outputs n = t + 64 e >= 255 only")
        for tau in range(2):
            G = self.bfly32([r[ke + 32 * tau] for ke in range(32)])
            for e in range(3, 32):
                voff = self.V_OFF3 if e == 3 else self.V_OFF
                self.emit("buffer_store_dword", None, [G[e].hi, ("r", voff)], imm=self.block_off(2 * tau, e))
                self.emit("buffer_store_dword", None, [G[e].lo, ("r", voff)], imm=self.block_off(2 * tau + 1, e))

    # the scheduler keeps LDS operations in order; stores have no ordering constraint beyond their data
    def build_unit(self, window=64):
        self.ins = []
        self.emit_unit()
        self.schedule(window)
        self.allocate()
        self.insert_waits()
        self.fix_permlane_hazards()
        body = self.ins
        loads = self.build_loads()
        body = self.merge_loads(body, loads)
        return self.insert_vm_waits(body)

    def insert_vm_waits(self, body):
        """Counted vmcnt waits instead of one vmcnt(0) at the top of a unit.  The unit's VMEM operations -- the NEXT unit's loads, spread through the body,
        and this unit's 116 stores at its end -- retire in issue order, so a register loaded early in the previous iteration needs no wait at all when it
        is first read, and only the loads issued among the stores make their first reader wait for the stores in front of them.  (One vmcnt(0) at the top
        made every unit wait for its predecessor's whole store burst.)  The loop is entered with everything retired (vmcnt(0) behind the prologue's
        loads), which only makes the counted waits of the first iteration easier to satisfy."""
        seq = [i for i in body if i.op in ("buffer_load_dword", "buffer_store_dword")]
        q = [set(keys_of(i.d)) if i.op == "buffer_load_dword" else None for i in seq]      # what the previous iteration left outstanding, oldest first
        out = []
        self.vm_waits = 0
        for i in body:
            regs = {key for o in i.operands() for key in keys_of(o)}
            need = -1
            for pos, dst in enumerate(q):
                if dst and dst & regs:
                    need = pos
            if need >= 0:
                cnt = min(len(q) - 1 - need, 62)
                out.append(Ins("s_waitcnt_vm", imm=cnt))
                self.vm_waits += 1
                if cnt < len(q):
                    del q[: len(q) - cnt]      # at most `cnt` operations are still outstanding: the oldest have retired
            out.append(i)
            if i.op == "buffer_load_dword":
                q.append(set(keys_of(i.d)))
            elif i.op == "buffer_store_dword":
                q.append(None)
        return out


def render(ins, g=OGen):
    def R(o):
        if o[0] == "v":
            return f"v[{o[1]}:{o[1] + 1}]"
        if o[0] == "h":
            return f"v{o[1] + o[2]}"
        if o[0] == "r":
            return f"v{o[1]}"
        return f"s[{o[1]}:{o[1] + 1}]"

    if ins.op in ("buffer_load_dword", "buffer_store_dword"):
        k, imm = divmod(ins.imm, 4096)
        so = "0" if k == 0 else f"s{g.S_K4 + k - 1}"
        if ins.op == "buffer_load_dword":
            return f"buffer_load_dword {R(ins.d)}, {R(ins.s[0])}, s[{g.S_RX}:{g.S_RX + 3}], {so} offen offset:{imm}"
        return f"buffer_store_dword {R(ins.s[0])}, {R(ins.s[1])}, s[{g.S_RY}:{g.S_RY + 3}], {so} offen offset:{imm}"
    return W.render(ins, {})


class OEmu(W.Emu):
    def __init__(self, g, sconst):
        super().__init__(g, sconst)
        self.lds = np.zeros(LDS_BYTES // 4, dtype=np.float32)
        self.out = None
        self.obase = 0

    def run(self, ins_list):
        for ins in ins_list:
            if ins.op == "buffer_store_dword":
                voff = self.vi[self.reg(ins.s[1])]
                ok = (voff >= 0) & (voff < 4 * NFFT)
                addr = self.obase + ins.imm + voff
                self.out[addr[ok] // 4] = self.v[self.reg(ins.s[0])][ok]
            else:
                super().run([ins])


def check(window=64):
    rng = np.random.default_rng(1776)
    g = OGen()
    unit = g.build_unit(window)
    loads = g.build_loads()
    nunits = 2
    nx = (4 * (nunits + 2)) * L + 4096
    x = rng.standard_normal(nx).astype(np.float32)
    taps = (rng.standard_normal(NB) / np.sqrt(NB)).astype(np.float32)
    Hc = np.fft.fft(np.concatenate([taps.astype(np.float64), np.zeros(NFFT - NB)])) / NFFT
    em = OEmu(g, W.sconsts(g))
    lane = np.arange(64)
    em.vi[g.V_OFF] = lane * 4
    em.vi[g.V_OFF3] = np.where(lane == 63, lane * 4, -(2 ** 31))
    wave = 5
    xb = H_BYTES + wave * XBUF_BYTES
    em.vi[g.V_XW] = xb + ((lane >> 5) * 32 + (lane & 31)) * XROW * 8
    em.vi[g.V_XR] = xb + ((lane >> 5) * 32) * XROW * 8 + (lane & 31) * 8
    em.vi[g.V_H] = (lane & 31) * 8
    hl = em.lds[: H_BYTES // 4].reshape(NFFT, 2)
    hl[:, 0] = Hc.real.astype(np.float32)
    hl[:, 1] = Hc.imag.astype(np.float32)
    roots = np.exp(-2j * np.pi * np.arange(NFFT) / NFFT)
    ke = lane & 31
    for j in range(1, 8):
        wa = roots[(8 * ke * j) % NFFT]
        wb = roots[(ke * j) % NFFT]
        em.v[g.TW0 + 2 * (j - 1)] = wa.real.astype(np.float32)
        em.v[g.TW0 + 2 * (j - 1) + 1] = wa.imag.astype(np.float32)
        em.v[g.TW0 + 14 + 2 * (j - 1)] = wb.real.astype(np.float32)
        em.v[g.TW0 + 14 + 2 * (j - 1) + 1] = wb.imag.astype(np.float32)
    em.glob = x
    em.out = np.full(nx, np.nan, dtype=np.float32)
    g_first = 4
    em.v[g.POOL0:] = np.float32(np.nan)
    em.gbase = (g_first * L - LEAD) * 4
    em.run(loads)
    for u in range(nunits):
        g0 = g_first + 4 * u
        em.gbase = ((g0 + 4) * L - LEAD) * 4          # the body carries the NEXT unit's loads
        em.obase = (g0 * L - LEAD) * 4
        em.run(unit)
    ref = np.convolve(x.astype(np.float64), taps.astype(np.float64))[:nx]
    lo, hi = g_first * L, (g_first + 4 * nunits) * L
    got = em.out[lo:hi].astype(np.float64)
    assert not np.any(np.isnan(got)), f"{int(np.isnan(got).sum())} outputs were never stored"
    assert np.all(np.isnan(em.out[:lo])) and np.all(np.isnan(em.out[hi:])), "stores outside the units' outputs"
    err = np.linalg.norm(got - ref[lo:hi]) / np.linalg.norm(ref[lo:hi])
    kinds = {}
    for i in unit:
        kinds[i.op] = kinds.get(i.op, 0) + 1
    nins = sum(v for k, v in kinds.items() if k != "comment")
    nbad = W.verify_waits(loads + [Ins("s_waitcnt_vm", imm=0)] + unit + unit + unit)
    print(f"emulated {nunits} units (8 blocks): relerr {err:.3e}; wait check over three iterations: {nbad} problems, {g.vm_waits} counted vmcnt waits; {nins} instructions per unit ({nins / (4 * L):.3f} per sample), peak live pairs {g.maxlive} of {g.NPOOL}, "
          f"scheduler stall slots {g.stalls}, {g.loads_in_body} of 128 loads inside the body")
    print("  ", {k: v for k, v in sorted(kinds.items()) if k != "comment"})
    return err < 2e-6 and nbad == 0




def fbits(x):
    return "0x%08x" % int(np.float32(x).view(np.uint32))


ABLATE = set()      # --ablate loads,stores,lds,nt : timing experiments (results are garbage except for nt = nontemporal stores)


def keep(ins):
    if "loads" in ABLATE and ins.op == "buffer_load_dword":
        return False
    if "stores" in ABLATE and ins.op == "buffer_store_dword":
        return False
    if "lds" in ABLATE and ins.op in ("ds_read_b64", "ds_write_b64", "s_waitcnt_lgkm"):
        return False
    return True


def kernel_text():
    g = OGen()
    Ls = []
    A = Ls.append
    NAME = "mdsp_ols_w64_asm"
    A('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
    A("\t.amdhsa_code_object_version 6")
    A("\t.text")
    A(f"\t.protected\t{NAME}")
    A(f"\t.globl\t{NAME}")
    A("\t.p2align\t8")
    A(f"\t.type\t{NAME},@function")
    A(f"{NAME}:")
    A("; generated by tools/gen_ols_asm.py -- do not edit")
    # s[0:1] kernarg, s2 = workgroup x, s3 = workgroup y (column), v0 = thread id
    A("\ts_load_dwordx8 s[4:11], s[0:1], 0x0          ; x, y, H, tw")
    A("\ts_load_dwordx8 s[12:19], s[0:1], 0x20        ; ldx, ldy, g_first, nunits")
    A("\ts_load_dwordx2 s[20:21], s[0:1], 0x40        ; run_len")
    A("\tv_and_b32_e32 v1, 63, v0                     ; lane")
    A("\tv_lshrrev_b32_e32 v2, 6, v0                  ; wave")
    A("\ts_nop 1                                       ; gfx940+: a VALU write of a VGPR needs a wait state before v_readfirstlane reads it")
    A("\tv_readfirstlane_b32 s22, v2")
    A("\ts_nop 3")
    A("\ts_waitcnt lgkmcnt(0)")
    # filter spectrum: 16 KiB global -> LDS, 32 bytes per thread
    A("\tv_lshlrev_b32_e32 v3, 5, v0                  ; tid * 32")
    A("\tglobal_load_dwordx4 v[40:43], v3, s[8:9]")
    A("\tglobal_load_dwordx4 v[44:47], v3, s[8:9] offset:16")
    A("\tv_mul_u32_u24_e32 v4, 112, v1")
    for k in range(7):
        A(f"\tglobal_load_dwordx4 v[{OGen.TW0 + 4 * k}:{OGen.TW0 + 4 * k + 3}], v4, s[10:11] offset:{16 * k}")
    A("\ts_waitcnt vmcnt(7)")
    A("\tds_write_b128 v3, v[40:43]")
    A("\tds_write_b128 v3, v[44:47] offset:16")
    A("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    A("\ts_barrier")
    # addresses
    A(f"\tv_lshlrev_b32_e32 v{OGen.V_OFF}, 2, v1               ; lane * 4")
    A("\tv_lshrrev_b32_e32 v40, 5, v1                 ; lane >> 5")
    A("\tv_and_b32_e32 v41, 31, v1                    ; lane & 31")
    A(f"\ts_mul_i32 s23, s22, {XBUF_BYTES}")
    A(f"\ts_add_i32 s23, s23, {H_BYTES}                 ; this wave's exchange buffer")
    A("\tv_lshl_add_u32 v42, v40, 5, v41")
    A(f"\tv_mul_u32_u24_e32 v42, {XROW * 8}, v42")
    A(f"\tv_add_u32_e32 v{OGen.V_XW}, s23, v42")
    A(f"\tv_mul_u32_u24_e32 v42, {32 * XROW * 8}, v40")
    A("\tv_lshl_add_u32 v42, v41, 3, v42")
    A(f"\tv_add_u32_e32 v{OGen.V_XR}, s23, v42")
    A(f"\tv_lshlrev_b32_e32 v{OGen.V_H}, 3, v41                ; (lane & 31) * 8")
    A("\tv_mov_b32_e32 v42, 0x80000000")
    A("\tv_cmp_eq_u32_e32 vcc, 63, v1")
    A(f"\tv_cndmask_b32_e32 v{OGen.V_OFF3}, v42, v{OGen.V_OFF}, vcc     ; lane 63: its offset; the others: out of range (their element 3 lies in front of the valid outputs)")
    # constants
    for k in range(7):
        A(f"\ts_mov_b32 s{OGen.S_K4 + k}, {4096 * (k + 1)}")
    h = np.float32(math.sqrt(0.5))
    A(f"\ts_mov_b32 s{OGen.S_HH}, {fbits(h)}")
    A(f"\ts_mov_b32 s{OGen.S_HH + 1}, {fbits(h)}")
    A(f"\ts_mov_b32 s{OGen.S_PM}, {fbits(1.0)}")
    A(f"\ts_mov_b32 s{OGen.S_PM + 1}, {fbits(-1.0)}")
    for m, p in g.wexp.items():
        c, s_ = W.w64(m)
        A(f"\ts_mov_b32 s{p}, {fbits(c)}                   ; W64^{m}")
        A(f"\ts_mov_b32 s{p + 1}, {fbits(s_)}")
    # slot, units
    A("\ts_lshl_b32 s23, s2, 3")
    A("\ts_add_u32 s23, s23, s22                      ; slot")
    A("\ts_mul_i32 s88, s23, s20                      ; u0 = slot * run_len")
    A("\ts_add_u32 s89, s88, s20")
    A("\ts_min_u32 s89, s89, s18                      ; uend")
    A("\ts_cmp_ge_u32 s88, s89")
    A("\ts_cbranch_scc1 .Lend")
    A("\ts_sub_u32 s89, s89, s88                      ; units of this wave")
    # first block g0 = g_first + 4 u0; byte offset of its window / first output = g0 * 7172 - 1020
    A("\ts_lshl_b32 s90, s88, 2")
    A("\ts_add_u32 s90, s90, s16                      ; g0 (low 32 bits: block counts stay below 2^31)")
    A(f"\ts_mul_hi_u32 s91, s90, {L * 4}")
    A(f"\ts_mul_i32 s90, s90, {L * 4}")
    A(f"\ts_sub_u32 s90, s90, {LEAD * 4}")
    A("\ts_subb_u32 s91, s91, 0")
    # column offsets col * ldx * 4, col * ldy * 4
    A("\ts_mul_i32 s92, s3, s12")
    A("\ts_mul_hi_u32 s93, s3, s12")
    A("\ts_mul_i32 s94, s3, s13")
    A("\ts_add_u32 s93, s93, s94")
    A("\ts_lshl_b64 s[92:93], s[92:93], 2")
    A("\ts_add_u32 s24, s4, s92")
    A("\ts_addc_u32 s25, s5, s93")
    A("\ts_add_u32 s24, s24, s90")
    A("\ts_addc_u32 s25, s25, s91")
    A("\ts_and_b32 s25, s25, 0xffff")
    A(f"\ts_mov_b32 s26, {3 * L * 4 + NFFT * 4}")
    A("\ts_mov_b32 s27, 0x00020000")
    A("\ts_mul_i32 s92, s3, s14")
    A("\ts_mul_hi_u32 s93, s3, s14")
    A("\ts_mul_i32 s94, s3, s15")
    A("\ts_add_u32 s93, s93, s94")
    A("\ts_lshl_b64 s[92:93], s[92:93], 2")
    A("\ts_add_u32 s28, s6, s92")
    A("\ts_addc_u32 s29, s7, s93")
    A("\ts_add_u32 s28, s28, s90")
    A("\ts_addc_u32 s29, s29, s91")
    A("\ts_and_b32 s29, s29, 0xffff")
    A(f"\ts_mov_b32 s30, {3 * L * 4 + NFFT * 4}")
    A("\ts_mov_b32 s31, 0x00020000")
    unit = g.build_unit()
    loads = g.build_loads()
    for ins in loads:
        if keep(ins):
            A("\t" + render(ins))
    A("\ts_waitcnt vmcnt(0)")
    A(".Lunit:")
    A(f"\ts_add_u32 s24, s24, {UNIT_BYTES}                ; the loads inside the body fetch the NEXT unit ...")
    A("\ts_addc_u32 s25, s25, 0")
    A("\ts_cmp_eq_u32 s89, 1")
    A(f"\ts_cselect_b32 s26, 0, {3 * L * 4 + NFFT * 4}      ; ... which does not exist behind this wave's last one: an empty descriptor returns zeros")
    for ins in unit:
        if keep(ins):
            A("\t" + render(ins) + (" nt" if "nt" in ABLATE and ins.op == "buffer_store_dword" else ""))
    A(f"\ts_add_u32 s28, s28, {UNIT_BYTES}")
    A("\ts_addc_u32 s29, s29, 0")
    A("\ts_sub_u32 s89, s89, 1")
    A("\ts_cmp_lg_u32 s89, 0")
    A("\ts_cbranch_scc1 .Lunit")
    A(".Lend:")
    A("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    A("\ts_endpgm")
    A("\t.section\t.rodata,\"a\",@progbits")
    A("\t.p2align\t6, 0x0")
    A(f"\t.amdhsa_kernel {NAME}")
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
    A("\t\t.amdhsa_reserve_vcc 1")
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
    A(f"\t.size\t{NAME}, .Lfunc_end0-{NAME}")
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
    A(f"    .name:           {NAME}")
    A("    .private_segment_fixed_size: 0")
    A("    .sgpr_count:     104")
    A("    .sgpr_spill_count: 0")
    A(f"    .symbol:         {NAME}.kd")
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
    nbody = sum(1 for i in unit if i.op != "comment")
    return "\n".join(Ls) + "\n", nbody


if __name__ == "__main__":
    if "--check" in sys.argv:
        sys.exit(0 if check() else 1)
    if "--ablate" in sys.argv:
        ABLATE.update(sys.argv[sys.argv.index("--ablate") + 1].split(","))
    text, nbody = kernel_text()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp.jl_amd", "csrc", "ols_w64_asm.s")
    open(out, "w").write(text)
    print(f"wrote {out}: {text.count(chr(10))} lines, {nbody} instructions per unit")
