#!/usr/bin/env python3
"""Generator (and CPU emulator) of `mdsp_welch_w64c_asm`: the hand-allocated Welch kernel of tools/gen_welch_asm.py with the shared half-frame CARRIED.

A unit is a frame pair = three half-frames (H0, H1, H2), and H2 is the next unit's H0.  `mdsp_welch_w64_asm` had no register left to keep it and loaded
it again one unit later: 96 loads per unit instead of 64, and 1.32 x the algorithmic bytes from HBM (profiles/r04y_pmc_traffic.json: most of the
re-reads miss the 4 MiB L2 of an XCD that has streamed 4 MiB of other waves' samples in between).  Here:
  * the 28 per-lane twiddle values move from registers into LDS (14 rows of 512 bytes in front of the window pairs, written once per workgroup, read
    back as 14 ds_read_b64 per unit), which frees 14 register pairs;
  * a unit's H0 and H2 values live COMPACTLY, one 32-bit register each, in two banks of 32 registers (16 pairs): bank C holds the carried half, bank N
    receives the new one and IS the next unit's bank C -- nothing is moved; the first layer's product (H0 w_lo, H2 w_hi) becomes two v_mul_f32 that take
    their operands from the two banks (one v_pk_mul_f32 before: +32 instructions per unit);
  * bank N is locked for the whole unit (16 pairs never enter the pool): 94 - 16 = 78 pairs for a peak of 72;
  * the roles of the banks alternate, so the loop body is TWO units (A: C = bank 0, B: C = bank 1), each scheduled and allocated on its own.

    python tools/gen_welch_asm_c.py [--check] [--ablate loads,lds,perm,acc]      writes dsp.jl_amd/csrc/welch_w64c_asm.s
Kernel contract: gen_welch_asm.py's (same W64AsmArgs, same partial rows, grid, flush rule); LDS: twiddles 7 KiB + window pairs 16 KiB + 8 x 16.5 KiB.
"""
import math
import os
import sys
from collections import deque

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_welch_asm as W
from gen_welch_asm import HALF, N, XBUF_BYTES, XROW, FLUSH, Ins, Val, VBASE, keys_of, slot64

TW_BYTES = 14 * 512
WIN_OFF = TW_BYTES
XB_OFF = TW_BYTES + W.WIN_BYTES
LDS_BYTES = XB_OFF + 8 * XBUF_BYTES
NAME = "mdsp_welch_w64c_asm"


def render(ins, pmap):
    if ins.op == "v_mul_f32":
        def R(o):
            p = o[1]
            if o[0] == "h":
                return f"v{(pmap[p] if p >= VBASE else p) + o[2]}"
            return f"v{p}"
        return f"v_mul_f32_e32 {R(ins.d)}, {R(ins.s[0])}, {R(ins.s[1])}"
    return W.render(ins, pmap)


class GenC(W.Gen):
    POOL0 = 68
    NPOOL = 94
    BANK = (68, 100)     # two banks of 16 pairs: value e of a half-frame's 32 per lane sits in 32-bit register BANK[b] + e
    HH0 = 132            # H1 as pairs, as in gen_welch_asm.py

    def __init__(self, parity):
        super().__init__()
        self.parity = parity

    def cval(self, bank, e):
        return ("h", self.BANK[bank] + 2 * (e // 2), e % 2)

    def hh_layout(self):
        return [[Val(self.HH0 + 2 * (2 * n1 + q)) for q in range(2)] for n1 in range(8)]

    def bank_pairs(self, b):
        return {self.BANK[b] + 2 * j for j in range(16)}

    def input_pairs(self):
        return self.bank_pairs(0) | self.bank_pairs(1) | {v.p for g in self.hh_layout() for v in g}

    # ---- loads: first = the prologue's (all three half-frames of the first unit), else the NEXT unit's H1 and H2 issued inside this unit's body:
    # its H2 goes into THIS unit's bank C (free behind the first layer), this unit's bank N is its bank C already
    def emit_loads(self, first=False):
        hh = self.hh_layout()
        c, n = self.parity, 1 - self.parity
        dst_n = n if first else c
        if first:
            for e in range(32):
                self.buffer_load(self.cval(c, e), 256 * e)
        for n1 in range(8):
            for j in range(4):
                e = n1 + 8 * j
                self.buffer_load(self.cval(dst_n, e), 2 * HALF * 4 + 256 * e)
            for q in range(2):
                self.buffer_load(hh[n1][q].lo, HALF * 4 + 256 * (n1 + 8 * (2 * q)))
                self.buffer_load(hh[n1][q].hi, HALF * 4 + 256 * (n1 + 8 * (2 * q + 1)))

    def emit_unit(self):
        hh = self.hh_layout()
        c, n = self.parity, 1 - self.parity
        v = [None] * 64
        self.comment("pass A, first layer (window folded in) + first radix-8 layer; H0 from the carried bank, H2 from the new one")
        for n1 in range(8):
            wp = [self.ds_read(self.V_WIN, WIN_OFF + 512 * (n1 + 8 * j)) for j in range(4)]
            S, D = [], []
            for j in range(4):
                e = n1 + 8 * j
                h = j & 1
                T = self.alloc()
                self.emit("v_mul_f32", ("h", T.p, 0), [self.cval(c, e), ("h", wp[j].p, 0)])
                self.emit("v_mul_f32", ("h", T.p, 1), [self.cval(n, e), ("h", wp[j].p, 1)])
                s_ = self.pk("v_pk_fma_f32", [hh[n1][j >> 1], wp[j], T], {"op_sel": [h, 1, 0], "op_sel_hi": [h, 0, 1]})
                d_ = self.pk("v_pk_fma_f32", [hh[n1][j >> 1], wp[j], T],
                             {"op_sel": [h, 1, 0], "op_sel_hi": [h, 0, 1], "neg_lo": [1, 0, 0], "neg_hi": [0, 0, 1]}, dst=T)
                S.append(s_)
                D.append(d_)
            o = self.bfly8_sd(S, D)
            for k1 in range(8):
                v[n1 + 8 * k1] = o[k1]
        self.comment("pass A, second radix-8 layer; half exchange by lane swaps; round 0 of the 64 x 64 transposition")
        m = [None] * 64
        for k1 in range(8):
            u = [v[j + 8 * k1] for j in range(8)]
            u = [u[0]] + [self.mul_w64(u[j], (j * k1) & 63) for j in range(1, 8)]
            out = self.bfly8(u)
            for k2 in range(4):
                lo_, hi_ = out[k2], out[k2 + 4]
                for half in ("lo", "hi"):
                    self.emit("v_permlane32_swap", None, [getattr(lo_, half), getattr(hi_, half)])
                m[k1 + 8 * k2] = lo_
                m[k1 + 8 * (k2 + 4)] = hi_
            for k2 in range(4):
                self.ds_write(self.V_XW, 8 * (k1 + 8 * k2), m[k1 + 8 * k2])
        nv = [None] * 64
        order = sorted(range(32), key=lambda T: (T & 7, T >> 3))
        for T in order:
            nv[T] = self.ds_read(self.V_XR, 8 * XROW * T)
        for r in range(32):
            self.ds_write(self.V_XW, 8 * r, m[32 + r])
        for T in order:
            nv[32 + T] = self.ds_read(self.V_XR, 8 * XROW * T)
        v = nv
        self.comment("pass B: the two-level twiddles come from LDS (row j - 1: W^(8 lane j), row 6 + j: W^(lane j)), one read per value and unit")
        tw = {}
        for t2 in (1, 2, 3, 4, 5, 6, 7):
            wa = self.ds_read(self.V_WIN, 512 * (t2 - 1))
            for t1 in range(8):
                tw[(t1, t2)] = self.cmul(v[t1 + 8 * t2], wa)
        wb = {t1: self.ds_read(self.V_WIN, 512 * (7 + t1 - 1)) for t1 in range(1, 8)}
        for t1 in range(8):
            q = [v[t1]] + [tw[(t1, t2)] for t2 in range(1, 8)]
            out = self.bfly8(q)
            for k1 in range(8):
                v[t1 + 8 * k1] = out[k1] if t1 == 0 else self.cmul(out[k1], wb[t1])
        self.bfly64_tail(v)
        self.comment("power: acc[s] += re^2 + im^2")
        for s in range(64):
            a = ("r", self.ACC0 + s)
            self.emit("v_fma_f32", a, [v[s].lo, v[s].lo, a])
            self.emit("v_fma_f32", a, [v[s].hi, v[s].hi, a])

    # ---- register allocation: as gen_welch_asm.py, with this unit's bank N never released
    def allocate(self):
        last = {}
        for k, i in enumerate(self.ins):
            for o in i.operands():
                if o[0] in ("v", "h"):
                    last[o[1]] = k
        inputs = self.input_pairs()
        locked = self.bank_pairs(1 - self.parity)
        free = [p for p in range(self.POOL0, self.POOL0 + 2 * self.NPOOL, 2) if p not in inputs]
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
                    late = [q_ for q_ in free if q_ not in inputs]
                    p = min(late) if late else min(free)
                    free.remove(p)
                    pmap[o[1]] = p
                    live += 1
                    self.maxlive = max(self.maxlive, live)
            for o in i.operands():
                if o[0] in ("v", "h") and last.get(o[1]) == k:
                    phys = pmap[o[1]] if o[1] >= VBASE else o[1]
                    last[o[1]] = -1
                    if phys in locked:
                        continue
                    if self.POOL0 <= phys < self.POOL0 + 2 * self.NPOOL:
                        quar.append((k + 3, phys))
                        live -= 1

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

    def build_unit(self, window=64):
        self.ins = []
        self.emit_unit()
        self.schedule(window)
        self.allocate()
        self.insert_waits()
        self.fix_permlane_hazards()
        body = self.ins
        self.ins = []
        self.emit_loads(first=False)
        loads = [i for i in self.ins]
        # a load into this unit's bank C may only follow the last use of its PAIR in the body; merge_loads keys on the pair of the destination
        body = self.merge_loads(body, loads)
        return [Ins("s_waitcnt_vm", imm=0)] + body

    def build_first_loads(self):
        self.ins = []
        self.emit_loads(first=True)
        return [i for i in self.ins]


class EmuC(W.Emu):
    def run(self, ins_list):
        for ins in ins_list:
            if ins.op == "v_mul_f32":
                a, b = (self.v[self.reg(s)] for s in ins.s)
                self.v[self.reg(ins.d)] = (a * b).astype(np.float32)
            else:
                super().run([ins])


def build():
    gA, gB = GenC(0), GenC(1)
    return gA, gB, gA.build_unit(), gB.build_unit(), GenC(0).build_first_loads()


def check():
    rng = np.random.default_rng(1776)
    gA, gB, bodyA, bodyB, loads = build()
    nunits = 5
    sig = rng.standard_normal((nunits + 3) * N).astype(np.float32)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / (N - 1))).astype(np.float32)
    em = EmuC(gA, W.sconsts(gA))
    em.lds = np.zeros(LDS_BYTES // 4, dtype=np.float32)
    lane = np.arange(64)
    em.vi[gA.V_OFF] = lane * 4
    em.vi[gA.V_WIN] = lane * 8
    wave = 5
    xb = XB_OFF + wave * XBUF_BYTES
    em.vi[gA.V_XW] = xb + ((lane >> 5) * 32 + (lane & 31)) * XROW * 8
    em.vi[gA.V_XR] = xb + ((lane >> 5) * 32) * XROW * 8 + (lane & 31) * 8
    wl = em.lds[WIN_OFF // 4: (WIN_OFF + W.WIN_BYTES) // 4].reshape(HALF, 2)
    wl[:, 0] = win[:HALF]
    wl[:, 1] = win[HALF:]
    roots = np.exp(-2j * np.pi * np.arange(N) / N)
    twl = em.lds[: TW_BYTES // 4].reshape(14, 64, 2)
    for j in range(1, 8):
        wa, wb = roots[(8 * lane * j) % N], roots[(lane * j) % N]
        twl[j - 1, :, 0], twl[j - 1, :, 1] = wa.real, wa.imag
        twl[7 + j - 1, :, 0], twl[7 + j - 1, :, 1] = wb.real, wb.imag
    em.glob = sig
    ref = np.zeros(N)
    em.v[gA.POOL0:] = np.float32(np.nan)
    em.gbase = 0
    em.run(loads)
    for u in range(nunits):
        em.gbase = (u + 1) * N * 4
        em.run(bodyA if u % 2 == 0 else bodyB)
        a = sig[u * N: u * N + N].astype(np.float64)
        b = sig[u * N + HALF: u * N + HALF + N].astype(np.float64)
        ref += np.abs(np.fft.fft(win.astype(np.float64) * (a + 1j * b))) ** 2
    got = np.zeros(N)
    for kt in range(64):
        got[lane + 64 * kt] = em.v[gA.ACC0 + slot64(kt)]
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    worst = np.max(np.abs(got - ref) / ref.max())
    kinds = {}
    for i in bodyA:
        kinds[i.op] = kinds.get(i.op, 0) + 1
    nins = sum(v for k, v in kinds.items() if k != "comment")
    nbad = W.verify_waits(loads + bodyA + bodyB + bodyA + bodyB)
    print(f"emulated {nunits} units (A B A B A): relerr {err:.3e}, worst bin / max {worst:.3e}; {nins} instructions per unit, peak live pairs {gA.maxlive} / {gB.maxlive} of "
          f"{gA.NPOOL}, scheduler stall slots {gA.stalls} / {gB.stalls}, {gA.loads_in_body} / {gB.loads_in_body} of 64 loads inside the body, wait check: {nbad} problems")
    print("  ", {k: v for k, v in sorted(kinds.items()) if k != "comment"})
    return err < 2e-6 and nbad == 0


def kernel_text():
    gA, gB, bodyA, bodyB, loads = build()
    G = GenC
    L = []
    A = L.append
    A('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
    A("\t.amdhsa_code_object_version 6")
    A("\t.text")
    A(f"\t.protected\t{NAME}")
    A(f"\t.globl\t{NAME}")
    A("\t.p2align\t8")
    A(f"\t.type\t{NAME},@function")
    A(f"{NAME}:")
    A("; generated by tools/gen_welch_asm_c.py -- do not edit")
    A("\ts_load_dwordx8 s[4:11], s[0:1], 0x0          ; s, part, winpairs, tw")
    A("\ts_load_dwordx8 s[12:19], s[0:1], 0x20        ; lds_, units, run_len, nch")
    A("\ts_load_dword s20, s[0:1], 0x40               ; nflush")
    A("\tv_and_b32_e32 v1, 63, v0                     ; lane")
    A("\tv_lshrrev_b32_e32 v2, 6, v0                  ; wave")
    A("\ts_nop 1                                       ; gfx940+: a VALU write of a VGPR needs a wait state before v_readfirstlane reads it")
    A("\tv_readfirstlane_b32 s21, v2")
    A("\ts_nop 3")
    A("\ts_waitcnt lgkmcnt(0)")
    A("\tv_lshlrev_b32_e32 v3, 5, v0                  ; tid * 32: window pairs, 16 KiB global -> LDS behind the twiddle rows")
    A("\tglobal_load_dwordx4 v[8:11], v3, s[8:9]")
    A("\tglobal_load_dwordx4 v[12:15], v3, s[8:9] offset:16")
    A("\tv_mul_u32_u24_e32 v4, 112, v1                ; per-lane twiddles: 28 floats at tw + lane * 112")
    for k in range(7):
        A(f"\tglobal_load_dwordx4 v[{G.POOL0 + 4 * k}:{G.POOL0 + 4 * k + 3}], v4, s[10:11] offset:{16 * k}")
    A("\tv_lshlrev_b32_e32 v7, 3, v1                  ; lane * 8")
    A("\ts_waitcnt vmcnt(7)")
    A(f"\tds_write_b128 v3, v[8:11] offset:{WIN_OFF}")
    A(f"\tds_write_b128 v3, v[12:15] offset:{WIN_OFF + 16}")
    A("\ts_waitcnt vmcnt(0)")
    for k in range(14):
        A(f"\tds_write_b64 v7, v[{G.POOL0 + 2 * k}:{G.POOL0 + 2 * k + 1}] offset:{512 * k}      ; every wave writes the same 14 rows")
    A("\ts_waitcnt lgkmcnt(0)")
    A("\ts_barrier")
    A(f"\tv_lshlrev_b32_e32 v{G.V_OFF}, 2, v1               ; lane * 4")
    A("\tv_lshrrev_b32_e32 v5, 5, v1                  ; lane >> 5")
    A("\tv_and_b32_e32 v6, 31, v1                     ; lane & 31")
    A(f"\ts_mul_i32 s22, s21, {XBUF_BYTES}")
    A(f"\ts_add_i32 s22, s22, {XB_OFF}                ; this wave's exchange buffer")
    A("\tv_lshl_add_u32 v7, v5, 5, v6                 ; (lane >> 5) * 32 + (lane & 31)")
    A(f"\tv_mul_u32_u24_e32 v7, {XROW * 8}, v7")
    A(f"\tv_add_u32_e32 v{G.V_XW}, s22, v7")
    A(f"\tv_mul_u32_u24_e32 v7, {32 * XROW * 8}, v5")
    A("\tv_lshl_add_u32 v7, v6, 3, v7")
    A(f"\tv_add_u32_e32 v{G.V_XR}, s22, v7")
    A(f"\tv_lshlrev_b32_e32 v{G.V_WIN}, 3, v1              ; lane * 8   (v1 was the lane: last use above)")
    for s in range(64):
        A(f"\tv_mov_b32_e32 v{G.ACC0 + s}, 0")
    for k in range(5):
        A(f"\ts_mov_b32 s{G.S_K4 + k}, {4096 * (k + 1)}")
    h = np.float32(math.sqrt(0.5))

    def fbits(x):
        return "0x%08x" % int(np.float32(x).view(np.uint32))

    A(f"\ts_mov_b32 s{G.S_HH}, {fbits(h)}")
    A(f"\ts_mov_b32 s{G.S_HH + 1}, {fbits(h)}")
    A(f"\ts_mov_b32 s{G.S_PM}, {fbits(1.0)}")
    A(f"\ts_mov_b32 s{G.S_PM + 1}, {fbits(-1.0)}")
    for m, p in gA.wexp.items():
        c, s_ = W.w64(m)
        A(f"\ts_mov_b32 s{p}, {fbits(c)}                   ; W64^{m}")
        A(f"\ts_mov_b32 s{p + 1}, {fbits(s_)}")
    A("\ts_lshl_b32 s22, s2, 3")
    A("\ts_add_u32 s22, s22, s21                      ; slot")
    A("\ts_mul_i32 s23, s22, s16                      ; u0 = slot * run_len")
    A("\ts_add_u32 s86, s23, s16")
    A("\ts_min_u32 s86, s86, s14                      ; uend")
    A("\ts_cmp_ge_u32 s23, s86")
    A("\ts_cbranch_scc1 .Lend")
    A("\ts_sub_u32 s87, s86, s23                      ; units of this wave")
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
    for ins in loads:
        if W.keep(ins):
            A("\t" + render(ins, {}))

    def flush():
        for kt in range(64):
            k, imm = divmod(256 * kt, 4096)
            so = "0" if k == 0 else f"s{G.S_K4 + k - 1}"
            A(f"\tbuffer_store_dword v{G.ACC0 + slot64(kt)}, v{G.V_OFF}, s[28:31], {so} offen offset:{imm}")
        A("\ts_add_u32 s28, s28, 0x4000")
        A("\ts_addc_u32 s29, s29, 0")
        A("\ts_nop 4")
        for s in range(64):
            A(f"\tv_mov_b32_e32 v{G.ACC0 + s}, 0")
        A(f"\ts_mov_b32 s91, {FLUSH}")

    for tag, other, body in (("A", "B", bodyA), ("B", "A", bodyB)):
        A(f".Lunit{tag}:")
        A("\ts_waitcnt vmcnt(0)")
        A("\ts_add_u32 s24, s24, 0x4000                    ; the loads inside the body fetch the NEXT unit ...")
        A("\ts_addc_u32 s25, s25, 0")
        A("\ts_cmp_eq_u32 s87, 1")
        A("\ts_cselect_b32 s26, 0, 0x6000                  ; ... which does not exist behind this wave's last one: an empty descriptor returns zeros")
        for ins in body[1:]:
            if W.keep(ins):
                A("\t" + render(ins, {}))
        A("\ts_sub_u32 s87, s87, 1")
        A("\ts_cmp_eq_u32 s87, 0")
        A(f"\ts_cbranch_scc1 .Lflush{tag}")
        A("\ts_sub_u32 s91, s91, 1")
        A("\ts_cmp_lg_u32 s91, 0")
        A(f"\ts_cbranch_scc1 .Lunit{other}")
        A(f".Lflush{tag}:")
        A("; one row of Float32 sums: bin lane + 64 kt sits in accumulator slot64(kt)")
        flush()
        A("\ts_cmp_lg_u32 s87, 0")
        A(f"\ts_cbranch_scc1 .Lunit{other}")
        if tag == "A":
            A("\ts_branch .Lend")
    A(".Lend:")
    A("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    A("\ts_endpgm")
    A("\t.section\t.rodata,\"a\",@progbits")
    A("\t.p2align\t6, 0x0")
    A(f"\t.amdhsa_kernel {NAME}")
    A(f"\t\t.amdhsa_group_segment_fixed_size {LDS_BYTES}")
    for line in ("private_segment_fixed_size 0", "kernarg_size 72", "user_sgpr_count 2", "user_sgpr_dispatch_ptr 0", "user_sgpr_queue_ptr 0",
                 "user_sgpr_kernarg_segment_ptr 1", "user_sgpr_dispatch_id 0", "user_sgpr_kernarg_preload_length 0", "user_sgpr_kernarg_preload_offset 0",
                 "user_sgpr_private_segment_size 0", "uses_dynamic_stack 0", "enable_private_segment 0", "system_sgpr_workgroup_id_x 1",
                 "system_sgpr_workgroup_id_y 1", "system_sgpr_workgroup_id_z 0", "system_sgpr_workgroup_info 0", "system_vgpr_workitem_id 0",
                 "next_free_vgpr 256", "next_free_sgpr 96", "accum_offset 256", "reserve_vcc 0", "float_round_mode_32 0", "float_round_mode_16_64 0",
                 "float_denorm_mode_32 3", "float_denorm_mode_16_64 3", "dx10_clamp 1", "ieee_mode 1", "fp16_overflow 0", "tg_split 0"):
        A("\t\t.amdhsa_" + line)
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
    A("    .sgpr_count:     102")
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
    nbody = sum(1 for i in bodyA if i.op != "comment")
    return "\n".join(L) + "\n", nbody


if __name__ == "__main__":
    if "--check" in sys.argv:
        sys.exit(0 if check() else 1)
    if "--ablate" in sys.argv:
        W.ABLATE.update(sys.argv[sys.argv.index("--ablate") + 1].split(","))
    text, nbody = kernel_text()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp.jl_amd", "csrc", "welch_w64c_asm.s")
    open(out, "w").write(text)
    print(f"wrote {out}: {text.count(chr(10))} lines, {nbody} instructions per unit")
