#!/usr/bin/env python3
"""Generator of the hand-placed tile loop of k_conv_y.hip  ->  tools/variants/k_conv_y_gen.inc

k_conv_y.hip runs ONE wave per SIMD; every wave owns two 32-edge column blocks a / b.  A W2 tile is two HALF-BURSTS: HB(X = a) = the 28
MFMAs of block a (one accumulator chain), HB(X = b) the same for block b.  In the MFMA shadows of HB(X) - hand-placed (tools/probes/
mfma_probe14.hip: up to five single-issue instructions per v_mfma_f32_32x32x16_f16 are free for a lone wave; mfma_probe15.hip: the skeleton of this
loop runs 1940 cycles per tile against a floor of 1792) - rides the epilogue of the OTHER block Y, whose accumulator is complete: bias,
tensor-product FMAs, the flush of the column its previous tile closed (segmented DPP scan + atomics), plus the LDS fragment reads one K step
ahead and half of the ring traffic.

A half-burst is three segments:
  S1 (K steps 0,1; 12 MFMAs): fixed LDS / ring traffic, Y's bias, and the PENDING FLUSH of block Y    variants: NONE, S (scalar column), V (vector column), SKIP
  S2 (K steps 2,3; 12 MFMAs): the tile kind's tensor-product FMAs of block Y + the packed quad        variants: NONE, RA, RT, TV0..TV3, RTS
  S3 (tail, 4 MFMAs): fixed, at the end of S2's statement
S1 and S2 are ONE `asm volatile` statement each: the variant is selected by scalar branches INSIDE the statement, so that every variant works on the
same operand registers (a C++ switch around per-variant statements made the compiler copy 20-40 registers at every merge).  Values that are both
tuples (LDS read results) and read element-wise (bias, feature rows) live in fixed registers v0..v25, clobbered by the statements.
DRAIN statements (no MFMAs) close a unit.  This script only zips instruction streams and counts the LDS queue for the s_waitcnt lgkmcnt(N) in front
of every first use; the arithmetic is k_conv_x.hip's epilogue (reference models/tensor_layers.py:65-116,147-159).
"""
import os

LIMB = 4608
BIAS_OFF = 3 * LIMB

# fixed registers (transient inside one statement)
B = ['v0', 'v1', 'v2', 'v3', 'v4', 'v5', 'v6', 'v7', 'v8', 'v9', 'v10', 'v11', 'v12', 'v13', 'v14', 'v15']      # S1: bias b0..b3
F = B                                                                                                          # S2: f0, f1, f2, g0
FQ = [f'v{16 + i}' for i in range(9)]        # S1: flush values
FO = 'v25'                                   # S1: flush address
XQ = 'v16'                                   # S2: packed-quad dot product


def tup(regs):
    lo, hi = int(regs[0][1:]), int(regs[-1][1:])
    return f'v[{lo}:{hi}]'


class Emit:
    def __init__(self):
        self.lines = []
        self.T = {}
        self.lds_issued = 0
        self.lds_done = 0

    def raw(self, s):
        self.lines.append(s)

    def lds(self, name, stmt):
        self.lines.append(stmt)
        self.T[name] = self.lds_issued
        self.lds_issued += 1

    def need(self, name):
        if name not in self.T:
            raise RuntimeError(f'schedule error: {name} is used before its LDS read was issued')
        t = self.T[name]
        if t < self.lds_done:
            return
        self.lines.append(f's_waitcnt lgkmcnt({self.lds_issued - (t + 1)})')
        self.lds_done = t + 1

    def item(self, it):
        if isinstance(it, tuple) and it[0] == 'need':
            self.need(it[1])
        elif isinstance(it, tuple) and it[0] == 'lds':
            self.lds(it[1], it[2])
        else:
            self.raw(it)


def zip_streams(E, mfmas, fixed, streams, first_stream_cap=None):
    """mfmas: list of (needs, stmt); fixed: gap -> items behind MFMA i; streams: list of (first_gap, items), every stream spread evenly over the gaps
    [first_gap, n) - except stream `first_stream_cap[0]`, which is issued first_stream_cap[1] items per gap from its first gap on (needed early)"""
    n = len(mfmas)
    pos = [0] * len(streams)
    for i, (needs, stmt) in enumerate(mfmas):
        for nm in needs:
            E.need(nm)
        E.raw(stmt)
        for it in fixed.get(i, []):
            E.item(it)
        for si, (g0, items) in enumerate(streams):
            if i < g0:
                continue
            remaining = len(items) - pos[si]
            q = -(-remaining // (n - i)) if remaining > 0 else 0
            if first_stream_cap is not None and si == first_stream_cap[0]:
                q = min(remaining, first_stream_cap[1])
            for _ in range(q):
                E.item(items[pos[si]])
                pos[si] += 1
    for si, (g0, items) in enumerate(streams):
        assert pos[si] == len(items)


# ------------------------------------------------------------------------------------------------------------------------------------------
# operand names (asm side).  dy0..15 = D of block Y element-wise; aA / aV / aX / R = Y's accumulators
# ------------------------------------------------------------------------------------------------------------------------------------------
DBASE = {'a': 26, 'b': 42}       # Da = v[26:41], Db = v[42:57] (physical-register constraints in every statement that touches them)
AHM = 'v[58:61]'                 # {W_hi, W_mid} tail fragments; its halves feed the two K = 8 MFMAs
_Y = ['b']                       # block whose epilogue the current stream belongs to (set by the segment generators)


def dtup(blk):
    return f'v[{DBASE[blk]}:{DBASE[blk] + 15}]'


def dy(i):
    return f'v{DBASE[_Y[0]] + i}'


def aA(rq):
    return f'%[aA{rq}]'


def aV(rq, c):
    return f'%[aV{rq}{c}]'


def aX(rq, c):
    return f'%[aX{rq}{c}]'


def bias_stream():
    out = []
    for j in range(4):
        out.append(('need', f'b{j}'))
        for c in range(4):
            out.append(f'v_fmac_f32 {dy(4 * j + c)}, {B[4 * j + c]}, %[bsc2]')
    return out


def packed_stream():
    g = F[12:16]
    out = [('need', 'g0'), f'v_mul_f32 {XQ}, {g[3]}, {dy(15)}', f'v_fmac_f32 {XQ}, {g[2]}, {dy(14)}', f'v_fmac_f32 {XQ}, {g[1]}, {dy(13)}', f'v_fmac_f32 {XQ}, {g[0]}, {dy(12)}']
    for i in range(3):
        out.append(f'v_fmac_f32 {aA(i)}, %[pk{i}], {XQ}')
    return out


def main_stream(kind, packed=True):
    out = [('need', 'f0')]
    f = [F[0:4], F[4:8], F[8:12]]
    if kind in ('RA', 'RT'):
        acc = aA if kind == 'RA' else (lambda rq: aV(rq, 0))
        for j in (3, 2, 1, 0):          # fmaf(f.x, D0, fmaf(f.y, D1, fmaf(f.z, D2, fmaf(f.w, D3, acc))))
            for rq in range(4):
                out.append(f'v_fmac_f32 {acc(rq)}, {f[0][j]}, {dy(4 * rq + j)}')
        if packed:
            out += packed_stream()
    elif kind.startswith('TV'):
        x = int(kind[2])
        x01, x23 = bool(x & 1), bool(x & 2)
        for c in range(3):
            if c:
                out.append(('need', f'f{c}'))
            for j in (1, 0, 3, 2):
                cross = x01 if j < 2 else x23
                for rq in range(3):
                    tgt = aX(rq, c) if cross else aV(rq, c)
                    out.append(f'v_fmac_f32 {tgt}, {f[c][j]}, {dy(4 * rq + j)}')
            if c == 0 and packed:
                out += packed_stream()
    elif kind == 'RTS':
        for j in (1, 0):
            for rq in range(4):
                out.append(f'v_fmac_f32 {aV(rq, 0)}, {f[0][j]}, {dy(4 * rq + j)}')
        for rq in range(4):
            out.append(f'v_mul_f32 %[R{rq}], {f[0][3]}, {dy(4 * rq + 3)}')
        for rq in range(4):
            out.append(f'v_fmac_f32 %[R{rq}], {f[0][2]}, {dy(4 * rq + 2)}')
        if packed:
            out += packed_stream()
    return out


DPP = ['row_shr:1 row_mask:0xf bank_mask:0xf', 'row_shr:2 row_mask:0xf bank_mask:0xf', 'row_shr:4 row_mask:0xf bank_mask:0xf',
       'row_shr:8 row_mask:0xf bank_mask:0xf', 'row_bcast:15 row_mask:0xa bank_mask:0xf']
SM = ['sm1', 'sm2', 'sm4', 'sm8', 'sm16']


def scan_stream(vals):
    # a VALU result needs two wait states before a DPP read of it: the other channels' instructions provide them (>= 4 channels interleaved)
    out = []
    for step in range(5):
        for v in vals:
            out.append(f'v_fmac_f32_dpp {v}, {v}, %[{SM[step]}] {DPP[step]} bound_ctrl:1')
    return out


def atomics(vals, offsets):
    # ONE item: nothing else may be zipped between the two exec writes (run tails only)
    lines = ['s_mov_b64 %[sv], exec', 's_and_b64 exec, exec, %[tail]']
    for v, o in zip(vals, offsets):
        lines.append(f'global_atomic_add_f32 {FO}, {v}, %[sumbase]' + (f' offset:{o}' if o else ''))
    lines.append('s_mov_b64 exec, %[sv]')
    return ['\\n\\t'.join(lines)]


def flush_stream(mode):
    """the flush of the column block Y's previous tile closed (k_conv_x.hip finish_tile): out values, resets, segmented scan, atomics"""
    out = []
    if mode == 'S':
        m = FQ[0:4]
        for rq in range(4):
            out.append(f'v_fma_f32 {m[rq]}, {aA(rq)}, %[s0], {aV(rq, 0)}')
        for rq in range(4):
            out.append(f'v_mul_f32 {m[rq]}, %[oscv], {m[rq]}')
        for rq in range(4):
            out.append(f'v_mov_b32 {aA(rq)}, 0')
            out.append(f'v_mov_b32 {aV(rq, 0)}, %[R{rq}]')
            out.append(f'v_mov_b32 %[R{rq}], 0')
        out.append(f'v_add3_u32 {FO}, %[vrow], %[pchan], %[hh4]')
        out += scan_stream(m)
        out += atomics(m, [8 * rq for rq in range(4)])
    else:
        m = FQ[0:9]
        V = ['%[vx]', '%[vy]', '%[vz]']
        W = ['%[wx]', '%[wy]', '%[wz]']
        for rq in range(3):
            for c in range(3):          # osc * fma(X[c+1], w[c+2], fma(-X[c+2], w[c+1], fma(sa, v[c], s0 * accV[c])))
                c1, c2 = (c + 1) % 3, (c + 2) % 3
                t = m[3 * rq + c]
                out.append(f'v_mul_f32 {t}, %[s0], {aV(rq, c)}')
                out.append(f'v_fmac_f32 {t}, {aA(rq)}, {V[c]}')
                out.append(f'v_fma_f32 {t}, -{aX(rq, c2)}, {W[c1]}, {t}')
                out.append(f'v_fmac_f32 {t}, {aX(rq, c1)}, {W[c2]}')
        for t in m:
            out.append(f'v_mul_f32 {t}, %[oscv], {t}')
        for rq in range(3):
            out.append(f'v_mov_b32 {aA(rq)}, 0')
            for c in range(3):
                out.append(f'v_mov_b32 {aV(rq, c)}, 0')
                out.append(f'v_mov_b32 {aX(rq, c)}, 0')
        out.append(f'v_mov_b32 {aA(3)}, 0')
        out.append(f'v_add3_u32 {FO}, %[vrow], %[pchan], %[hh12]')
        out += scan_stream(m)
        out += atomics(m, [24 * rq + 4 * c for rq in range(3) for c in range(3)])
    return out


# ------------------------------------------------------------------------------------------------------------------------------------------
# segments.  One statement per HALF-BURST: [S1 dispatch + variants][S2 dispatch + variants (+ tail)][end].  Everything a half-burst needs that depends
# on the tile (descriptor decode, LDS addresses of the ring stage / feature rows) is computed in the MFMA shadows too - compiler-placed code between the
# statements (116 instructions per tile in the first version) stalls the matrix pipe: the wave has nothing queued behind the last MFMA.
# Scratch registers (fixed, clobbered): v62:65 alh, v66 fy, v67 gy, v68 na, v69 da, v70/v71 ring store addresses, v72:73 next descriptor.
# ------------------------------------------------------------------------------------------------------------------------------------------
ALH = 'v[62:65]'
VFY, VGY, VNA, VDA, VWA0, VWA1, VDQ = 'v66', 'v67', 'v68', 'v69', 'v70', 'v71', 'v[72:73]'
CLOBBER = [f'v{i}' for i in range(26)] + [f'v{i}' for i in range(58, 74)]
TB = 13968
DESC_OFF = BIAS_OFF + 128


def step_mfmas(X, s_local, set_, first=False, needs=()):
    """six limb products of one K step on the chain D_X (k_conv_x.hip's ONE_ACC order: the 2^-22 terms first); s_local: 0..3"""
    a = f'a{set_}'
    prods = [(f'{a}h', f'hl{s_local}'), (f'{a}l', f'hh{s_local}'), (f'{a}m', f'hm{s_local}'), (f'{a}h', f'hm{s_local}'), (f'{a}m', f'hh{s_local}'),
             (f'{a}h', f'hh{s_local}')]
    out = []
    for i, (fa, fb) in enumerate(prods):
        c = '0' if (first and i == 0) else dtup(X)
        out.append((list(needs) if i == 0 else [], f'v_mfma_f32_32x32x16_f16 {dtup(X)}, %[{fa}], %[{fb}], {c}'))
    return out


def dsr128(dst, addr, off):
    return f'ds_read_b128 {dst}, {addr}' + (f' offset:{off}' if off else '')


def J(*lines):
    """one stream item of several instructions that must stay adjacent (an s_cmp and its s_cselect: fillers in between may write SCC)"""
    return '\\n\\t'.join(lines)


def decode_stream(hb):
    """S1 shadows: the descriptor of Y's tile -> S2's selector, the packed-quad masks, the feature-row addresses, the ring store addresses"""
    out = [
        's_and_b32 %[t0], %[w], 3',
        's_bfe_u32 %[t1], %[w], 0x2000e',
        's_add_i32 %[t1], %[t1], 3',
        's_add_i32 %[t2], %[t0], 1',
        J('s_cmp_eq_u32 %[t0], 2', 's_cselect_b32 %[t2], %[t1], %[t2]'),
        J('s_cmp_eq_u32 %[t0], 3', 's_cselect_b32 %[sel2], 7, %[t2]'),
    ]
    if hb == 0:
        out.append(J('s_cmp_eq_u32 %[ti], 0', 's_cselect_b32 %[sel2], 0, %[sel2]'))      # a unit's first half-burst: no tile of block b yet
    out += [
        's_bfe_u32 %[t1], %[w], 0x20008',
        J('s_bitcmp1_b32 %[w], 7', 's_cselect_b32 %[t1], %[t1], 3'),
        J('s_cmp_eq_u32 %[t1], 0', 's_cselect_b32 %[pk0], 1.0, 0'),
        J('s_cmp_eq_u32 %[t1], 1', 's_cselect_b32 %[pk1], 1.0, 0'),
        J('s_cmp_eq_u32 %[t1], 2', 's_cselect_b32 %[pk2], 1.0, 0'),
        's_lshr_b32 %[t0], %[w], 14',
        's_and_b32 %[t0], %[t0], 0x3fc',
        f'v_add_u32 {VFY}, %[t0], %[frY]',
        's_lshr_b32 %[t0], %[w], 6',
        's_and_b32 %[t0], %[t0], 0xf0',
        f'v_add_u32 {VGY}, %[t0], %[frY]',
    ]
    return out


def ring_addr_stream(hb):
    """S1 gaps 0-3: the store addresses of this half-burst's two chunks (stage of tile i+2)"""
    return [
        's_add_i32 %[t3], %[ti], 2',
        's_and_b32 %[t3], %[t3], 3',
        f's_mul_i32 %[t3], %[t3], {TB}',
        f'v_add_u32 {VWA0}, %[t3], %[rw0]',
        f'v_add_u32 {VWA1}, %[t3], %[rw1]',
    ]


def next_stream(hb):
    """S2 shadows: state of the NEXT half-burst.  hb 0: bias address of tile i for both following epilogue slots, b's pending flush.
    hb 1: fragment addresses of tile i+1 (also this half-burst's prefetch address and the next descriptor's), a's pending flush."""
    out = []
    if hb == 0:
        out += [
            's_and_b32 %[t3], %[ti], 3',
            f's_mul_i32 %[t3], %[t3], {TB}',
            'v_add_u32 %[vea], %[t3], %[ringb]',
            's_bfe_u32 %[pend], %[w], 0x20002',
            's_lshl_b32 %[pchan], %[ch], 2',
        ]
    else:
        out += [
            's_bfe_u32 %[pend], %[w], 0x20002',
            's_lshl_b32 %[pchan], %[ch], 2',
        ]
    return out


def na_stream(hb):
    """S2 gaps 0-5 (before the prefetch of the next half-burst's first K step at gap 8)"""
    if hb == 0:
        return [f'v_mov_b32 {VNA}, %[vfa]']
    return [
        's_add_i32 %[t3], %[ti], 1',
        's_and_b32 %[t3], %[t3], 3',
        f's_mul_i32 %[t3], %[t3], {TB}',
        f'v_add_u32 {VNA}, %[t3], %[ringl]',
        f's_add_i32 %[t0], %[t3], %[ring0d]',
        f'v_mov_b32 {VDA}, %[t0]',
    ]


def gen_s1(X, Y, pend, hb):
    E = Emit()
    _Y[0] = Y
    mf = step_mfmas(X, 0, 0, first=True) + step_mfmas(X, 1, 1, needs=['a1l'])
    fa = '%[vfa]'
    fixed = {
        0: [('lds', 'a1h', dsr128('%[a1h]', fa, 1024)), ('lds', 'b0', dsr128(tup(B[0:4]), '%[vea]', BIAS_OFF))],
        1: [('lds', 'a1m', dsr128('%[a1m]', fa, LIMB + 1024)), ('lds', 'b1', dsr128(tup(B[4:8]), '%[vea]', BIAS_OFF + 16))],
        2: [('lds', 'a1l', dsr128('%[a1l]', fa, 2 * LIMB + 1024)), ('lds', 'b2', dsr128(tup(B[8:12]), '%[vea]', BIAS_OFF + 32))],
        3: [('lds', 'b3', dsr128(tup(B[12:16]), '%[vea]', BIAS_OFF + 48))],
        # the ring: this thread's two chunks of record i+2 (requested one tile ago, into AGPRs) into the stage tile i-2 has left; the request for record
        # i+3 follows in S2.  vmcnt(2): everything but the two requests of the previous half-burst has landed (ops retire in order; a flush's atomics sit
        # BEFORE its half-burst's requests).  Measured: the wait itself costs nothing (vmcnt(63): same time); ISSUING the four 1-KB requests per wave and
        # tile costs ~170 cycles per tile (no requests: 2283 -> 2115) - a lone wave has nothing behind the one MFMA in flight to cover a VMEM issue
        4: ['s_waitcnt vmcnt(2)', ('lds', 'w0', f'ds_write_b128 {VWA0}, %[c0]')],
        5: [('lds', 'w1', f'ds_write_b128 {VWA1}, %[c1]')],
        6: [('lds', 'a0h', dsr128('%[a0h]', fa, 2048))],
        7: [('lds', 'a0m', dsr128('%[a0m]', fa, LIMB + 2048))],
        8: [('lds', 'a0l', dsr128('%[a0l]', fa, 2 * LIMB + 2048))],
    }
    streams = [(0, ring_addr_stream(hb))]
    if pend in ('S', 'V'):
        streams.append((0, flush_stream(pend)))
    streams.append((1, decode_stream(hb)))
    if pend != 'SKIP':
        streams.append((4, bias_stream()))
    # the ring addresses are needed at gap 4: their five instructions go first
    E2 = Emit()
    zip_streams(E, mf, fixed, streams, first_stream_cap=(0, 2))
    E.raw('s_waitcnt lgkmcnt(0)')      # S2 starts on the step-2 fragments (the youngest LDS op was issued >= 3 MFMAs ago)
    return E.lines


def gen_s2(X, Y, kind, hb):
    E = Emit()
    _Y[0] = Y
    mf = step_mfmas(X, 2, 0) + step_mfmas(X, 3, 1, needs=['a1l'])
    fa = '%[vfa]'
    fixed = {
        0: [('lds', 'a1h', dsr128('%[a1h]', fa, 3072)), ('lds', 'f0', dsr128(tup(F[0:4]), VFY, 0))],
        1: [('lds', 'a1m', dsr128('%[a1m]', fa, LIMB + 3072)), ('lds', 'g0', dsr128(tup(F[12:16]), VGY, 0))],
        2: [('lds', 'a1l', dsr128('%[a1l]', fa, 2 * LIMB + 3072))],
        3: ['s_min_u32 %[t2], %[soff], %[soffmax]', 'buffer_load_dwordx4 %[c0], %[ck0], %[rsrc], %[t2] offen'],      # (requests past the unit's last tile re-read it)
        4: ['buffer_load_dwordx4 %[c1], %[ck1], %[rsrc], %[t2] offen'],
        6: [('lds', 'ahm', f'ds_read2st64_b64 {AHM}, %[vft] offset0:8 offset1:17')],
        7: [('lds', 'alh', f'ds_read2st64_b64 {ALH}, %[vft] offset0:26 offset1:8')],
        8: [('lds', 'n0h', dsr128('%[a0h]', VNA, 0))],
        9: [('lds', 'n0m', dsr128('%[a0m]', VNA, LIMB))],
        10: [('lds', 'n0l', dsr128('%[a0l]', VNA, 2 * LIMB))],
    }
    if hb == 1:
        fixed[11] = [('lds', 'dq', f'ds_read_b64 {VDQ}, {VDA}')]
    if kind.startswith('TV'):
        fixed[2].append(('lds', 'f1', dsr128(tup(F[4:8]), VFY, 16)))
        fixed[3].append(('lds', 'f2', dsr128(tup(F[8:12]), VFY, 32)))
    streams = [(0, na_stream(hb))]
    if kind != 'NONE':
        streams.append((2 if kind.startswith('TV') else 4, main_stream(kind)))
    streams.append((6, next_stream(hb)))
    zip_streams(E, mf, fixed, streams, first_stream_cap=(0, 2))
    # the tail (packed as in k_conv_x.hip) needs the two tail fragments; behind them in the queue: the next half-burst's three fragments (+ the descriptor)
    E.raw(f's_waitcnt lgkmcnt({4 if hb == 1 else 3})')
    E.raw(f'v_mfma_f32_32x32x16_f16 {dtup(X)}, {ALH}, %[thl], {dtup(X)}')
    E.raw(f'v_mfma_f32_32x32x16_f16 {dtup(X)}, {AHM}, %[tmh], {dtup(X)}')
    E.raw(f'v_mfma_f32_32x32x8_f16 {dtup(X)}, v[60:61], %[tmid], {dtup(X)}')
    E.raw(f'v_mfma_f32_32x32x8_f16 {dtup(X)}, v[58:59], %[thi], {dtup(X)}')
    return E.lines


def gen_end(hb):
    """behind the tail.  hb 1: the tile is over - fragment addresses of tile i+1 become current, the next descriptor, the ring request offset, the barrier"""
    if hb == 0:
        return []
    return [
        f'v_mov_b32 %[vfa], {VNA}',
        f'v_add_u32 %[vft], %[t3], %[ringt]',
        f's_add_u32 %[soff], %[soff], {TB}',
        's_mov_b32 %[wP], %[w]',
        's_mov_b32 %[chP], %[ch]',
        's_waitcnt lgkmcnt(0)',
        'v_readfirstlane_b32 %[w], v72',
        'v_readfirstlane_b32 %[ch], v73',
        's_barrier',
    ]


def gen_drain_main(Y, kind):
    E = Emit()
    _Y[0] = Y
    for j in range(4):
        E.lds(f'b{j}', dsr128(tup(B[4 * j:4 * j + 4]), '%[ea]', BIAS_OFF + 16 * j))
    E.raw('s_waitcnt lgkmcnt(0)')
    E.lds_done = E.lds_issued
    E.raw('s_nop 15')
    E.raw('s_nop 15')      # (the chain's last MFMA has just been issued: 8 passes + write-back before the VALU may read its result)
    E.raw('s_nop 7')
    for it in bias_stream():
        E.item(it)
    E.lds('f0', dsr128(tup(F[0:4]), '%[fy]', 0))
    E.lds('g0', dsr128(tup(F[12:16]), '%[gy]', 0))
    if kind.startswith('TV'):
        E.lds('f1', dsr128(tup(F[4:8]), '%[fy]', 16))
        E.lds('f2', dsr128(tup(F[8:12]), '%[fy]', 32))
    E.raw('s_waitcnt lgkmcnt(0)')
    E.lds_done = E.lds_issued
    for it in main_stream(kind):
        E.item(it)
    return E.lines


def gen_drain_flush(mode):
    E = Emit()
    for it in flush_stream(mode):
        E.item(it)
    return E.lines


def dispatch(variants, sel, tag):
    """variants: list of (selector value, lines), most frequent first; scalar branches; the LAST variant is the fall-through of the compare chain"""
    lines = []
    n = len(variants)
    for k, (val, _) in enumerate(variants[:-1]):
        lines.append(f's_cmp_eq_u32 %[{sel}], {val}')
        lines.append(f's_cbranch_scc1 .L{tag}v{k}_%=')
    order = [n - 1] + list(range(n - 1))
    for pos, k in enumerate(order):
        if k != n - 1:
            lines.append(f'.L{tag}v{k}_%=:')
        lines += variants[k][1]
        if pos != n - 1:
            lines.append(f's_branch .L{tag}end_%=')
    lines.append(f'.L{tag}end_%=:')
    return lines


def dispatch_split(variants, sel, tag):
    """variants: list of (selector value, lines), the COMMON one first.  Returns (inline, out_of_line): the common variant runs straight through one
    untaken branch; the others live behind the statement's exit jump and come back to the join label (a taken branch costs the lone wave a bubble the
    single queued MFMA does not cover: four of them per half-burst cost ~350 cycles per tile)"""
    common_val, common_body = variants[0]
    inline = [f's_cmp_lg_u32 %[{sel}], {common_val}', f's_cbranch_scc1 .L{tag}rare_%='] + common_body + [f'.L{tag}join_%=:']
    ool = [f'.L{tag}rare_%=:']
    rest = variants[1:]
    for k, (val, _) in enumerate(rest[:-1]):
        ool.append(f's_cmp_eq_u32 %[{sel}], {val}')
        ool.append(f's_cbranch_scc1 .L{tag}v{k}_%=')
    n = len(rest)
    order = [n - 1] + list(range(n - 1))
    for k in order:
        if k != n - 1:
            ool.append(f'.L{tag}v{k}_%=:')
        ool += rest[k][1]
        ool.append(f's_branch .L{tag}join_%=')
    return inline, ool


def asm_stmt(name, lines, outs, ins, clobbers):
    def op(lst):
        return ', '.join(f'[{n}] "{c}"({e})' for n, c, e in lst)
    body = ' \\\n    '.join('"' + ln + '\\n\\t"' for ln in lines)
    cl = ', '.join(f'"{c}"' for c in clobbers)
    return (f'#define {name}() \\\n  asm volatile( \\\n    {body} \\\n    : {op(outs)} \\\n    : {op(ins)} \\\n    : {cl})\n')


KIND_SEL = [('RA', 1), ('RT', 2), ('TV0', 3), ('TV1', 4), ('TV2', 5), ('TV3', 6), ('RTS', 7), ('NONE', 0)]
PEND_SEL = [('NONE', 0), ('S', 1), ('V', 2), ('SKIP', 3)]


def acc_ops(Y):
    ops = []
    for rq in range(4):
        ops.append((f'aA{rq}', '+v', f'accA{Y}[{rq}]'))
    for rq in range(4):
        for c in range(3):
            ops.append((f'aV{rq}{c}', '+v', f'accV{Y}[{rq}][{c}]'))
            ops.append((f'aX{rq}{c}', '+v', f'accX{Y}[{rq}][{c}]'))
    for rq in range(4):
        ops.append((f'R{rq}', '+v', f'R{Y}[{rq}]'))
    return ops


def frag_ops():
    return [(n, '+v', n) for n in ('a0h', 'a0m', 'a0l', 'a1h', 'a1m', 'a1l')]


def main():
    out = ['// GENERATED by tools/gen_conv_y.py - do not edit; the schedule (what rides in which MFMA shadow) lives in the generator.\n']
    for hb, (X, Y) in enumerate((('a', 'b'), ('b', 'a'))):
      for par in (0,):      # (two chunk register sets with requests two tiles ahead were measured: 2308 -> 2283 cycles per tile, not worth twice the code)
            c0, c1 = (f'c0{par}', f'c1{par}') if hb == 0 else (f'c2{par}', f'c3{par}')
            k0, k1 = ('ck0', 'ck1') if hb == 0 else ('ck2', 'ck3')
            r0, r1 = ('ringw0', 'ringw1') if hb == 0 else ('ringw2', 'ringw3')
            in1, ool1 = dispatch_split([(v, gen_s1(X, Y, p, hb)) for p, v in PEND_SEL], 'sel1', 'p')
            in2, ool2 = dispatch_split([(v, gen_s2(X, Y, k, hb)) for k, v in KIND_SEL], 'sel2', 'k')
            lines = in1 + in2 + gen_end(hb) + ['s_branch .Lexit_%='] + ool1 + ool2 + ['.Lexit_%=:']
            outs = [('DX', '+{' + dtup(X) + '}', f'D{X}'), ('DY', '+{' + dtup(Y) + '}', f'D{Y}')] + frag_ops() + acc_ops(Y)
            outs += [('c0', '+a', c0), ('c1', '+a', c1), ('vfa', '+v', 'vfa'), ('vft', '+v', 'vft'), ('vea', '+v', 'vea')]
            outs += [('pend', '+s', f'pend{Y}'), ('pchan', '+s', f'pchan{Y}')]
            if hb == 1:
                outs += [('w', '+s', 'wC'), ('ch', '+s', 'chC'), ('wP', '+s', 'wP'), ('chP', '+s', 'chP'), ('soff', '+s', 'soff')]
            outs += [(n, '=&s', f'{n}_') for n in ('t0', 't1', 't2', 't3', 'sel2', 'pk0', 'pk1', 'pk2', 'sv')]
            ins = [(f'hl{s}', 'a', f'H{X}.hl[{s}]') for s in range(4)] + [(f'hh{s}', 'a', f'H{X}.hh[{s}]') for s in range(4)] + [(f'hm{s}', 'a', f'H{X}.hm[{s}]') for s in range(4)]
            ins += [('thl', 'a', f'H{X}.thl'), ('tmh', 'a', f'H{X}.tmh'), ('tmid', 'a', f'H{X}.tmid'), ('thi', 'a', f'H{X}.thi')]
            ins += [(n, 'v', f'{n}{Y}') for n in ('bsc2', 'oscv', 's0', 'vx', 'vy', 'vz', 'wx', 'wy', 'wz', 'sm1', 'sm2', 'sm4', 'sm8', 'sm16', 'vrow')]
            ins += [('hh4', 'v', 'hh4'), ('hh12', 'v', 'hh12'), ('ringl', 'v', 'ringl'), ('ringt', 'v', 'ringt'), ('ringb', 'v', 'ringb'), ('rw0', 'v', r0), ('rw1', 'v', r1),
                    ('frY', 'v', f'fr{Y}'), ('ck0', 'v', k0), ('ck1', 'v', k1)]
            ins += [('rsrc', 's', 'rsrc'), ('sumbase', 's', 'sumbase'), ('tail', 's', f'tail{Y}'), ('sel1', 's', 'sel1_'), ('ti', 's', 'i'), ('ring0d', 's', 'ring0d'),
                    ('soffmax', 's', 'soffmax')]
            if hb == 0:
                ins += [('w', 's', 'wP'), ('ch', 's', 'chP'), ('soff', 's', 'soff')]
            out.append(asm_stmt(f'Y_HB_{X}{par}', lines, outs, ins, CLOBBER + ['memory', 'scc']))
    for Y in ('a', 'b'):
        lines = dispatch([(v, gen_drain_main(Y, k)) for k, v in KIND_SEL[:-1]], 'sel', 'k')
        outs = [('DY', '+{' + dtup(Y) + '}', f'D{Y}')] + acc_ops(Y)
        ins = [('bsc2', 'v', f'bsc2{Y}'), ('ea', 'v', 'ea'), ('fy', 'v', 'fy'), ('gy', 'v', 'gy'), ('pk0', 's', f'pk0{Y}'), ('pk1', 's', f'pk1{Y}'), ('pk2', 's', f'pk2{Y}'),
               ('sel', 's', 'sel_')]
        out.append(asm_stmt(f'Y_DRAIN_MAIN_{Y}', lines, outs, ins, CLOBBER + ['memory', 'scc']))
        lines = dispatch([(1, gen_drain_flush('S')), (2, gen_drain_flush('V'))], 'sel', 'f')
        outs = acc_ops(Y) + [('sv', '=&s', 'sv_')]
        ins = [(n, 'v', f'{n}{Y}') for n in ('oscv', 's0', 'vx', 'vy', 'vz', 'wx', 'wy', 'wz', 'sm1', 'sm2', 'sm4', 'sm8', 'sm16', 'vrow')]
        ins += [('hh4', 'v', 'hh4'), ('hh12', 'v', 'hh12'), ('pchan', 's', f'pchan{Y}'), ('tail', 's', f'tail{Y}'), ('sumbase', 's', 'sumbase'), ('sel', 's', 'sel_')]
        out.append(asm_stmt(f'Y_DRAIN_FLUSH_{Y}', lines, outs, ins, CLOBBER + ['memory', 'scc']))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'k_conv_y_gen.inc')
    with open(path, 'w') as f:
        f.write('\n'.join(out))
    print('wrote', os.path.normpath(path), sum(s.count('\n') for s in out), 'lines')


if __name__ == '__main__':
    main()
