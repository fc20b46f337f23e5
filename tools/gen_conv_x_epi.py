#!/usr/bin/env python3
"""Generator of k_conv_x.hip's tile epilogue as ONE `asm volatile` statement per ring stage  ->  disco_diffdock_amd/csrc/k_conv_x_epi_gen.inc   (round 5)

k_conv_x.hip alternates the two waves of a SIMD: one bursts 28 MFMAs while its partner runs the VALU epilogue of its tile; the tile period is the wave-serial
chain burst + epilogue + barrier, and the compiler's epilogue carries ~40 instructions that are not the algorithm's (23-27 v_mov at the merge of the fast and
the slow path, v_cndmask / v_cmp pairs, conservative lgkmcnt(0) waits: VERDICT r04 #1a).  This statement is the whole epilogue - bias, ring stores, fold,
tile kind, packed quad, flush - with the accumulators as in/out operands, D0 / D1 / f0 in pinned registers (element access) and scalar branches inside, so every
tile kind works on the same registers.  The instruction streams are tools/gen_conv_y.py's (same arithmetic, same order as the C++ epilogue it replaces;
reference models/tensor_layers.py:65-116,147-159)."""
import os
import gen_conv_y as G

TB, LIMB = 13968, 4608
DEFER = os.environ.get('GEN_DEFER', '0') != '0'      # 1: group A's flush one tile late (built and measured: no gain, profiles/r05_conv_kernel_experiments.md); default: both half-groups flush in the closing tile's epilogue
BIAS_OFF = 3 * LIMB
# registers: D0 = v[16:31] (in/out: folded in place), D1 = v[32:47], f0 = v[48:51]; scratch v0..v15 (bias; then f1 v0-3, f2 v4-7, g0 v8-11, R v12-15; flush values v0-8,
# w = v / sqrt2 in v9-11), v52 packed-quad sum / g0 address, v53 flush address
G.DBASE['x'] = 16
G._Y[0] = 'x'
G.B[:] = [f'v{i}' for i in range(16)]
F0 = [f'v{48 + i}' for i in range(4)]
G.F = F0 + [f'v{i}' for i in range(12)]
G.FQ[:] = [f'v{i}' for i in range(9)]
G.FO = 'v53'
G.XQ = 'v52'
R = ['v12', 'v13', 'v14', 'v15']
CLOBBER = [f'v{i}' for i in range(16)] + ['v52', 'v53']


def fix(lines):
    out = []
    for ln in lines:
        if isinstance(ln, tuple):
            continue          # ('need', ...) markers: the waits are placed by hand below
        for k in range(4):
            ln = ln.replace(f'%[R{k}]', R[k])
        ln = ln.replace('%[pchan]', '%[chan4]')
        out.append(ln)
    return out


def flush(mode, rts):
    G.FO, G.XQ = 'v53', 'v52'
    lines = fix(G.flush_stream(mode))
    if mode == 'S':
        res = []
        for ln in lines:
            if ln.startswith('v_mov_b32 v1') and ln.endswith(', 0') and ln.split()[1].rstrip(',') in R:
                continue                                              # (R is scratch here: no reset)
            if not rts:
                for k in range(4):
                    ln = ln.replace(f'v_mov_b32 %[aV{k}0], {R[k]}', f'v_mov_b32 %[aV{k}0], 0')
            res.append(ln)
        return res
    pre = ['v_mul_f32 v9, 0x3f3504f3, %[vx]', 'v_mul_f32 v10, 0x3f3504f3, %[vy]', 'v_mul_f32 v11, 0x3f3504f3, %[vz]']
    return pre + [ln.replace('%[wx]', 'v9').replace('%[wy]', 'v10').replace('%[wz]', 'v11') for ln in lines]


PEND = [f'%[P{i}]' for i in range(9)]


def flush_split(mode, rts):
    """(phase 1, phase 2) of a flush whose values live in the pending registers P0..P8: phase 1 = the column's values and the accumulator resets, phase 2 = address,
    segmented scan, atomics (on the column offset kept in pchan)"""
    keep = list(G.FQ)
    G.FQ[:] = PEND
    lines = flush(mode, rts)
    G.FQ[:] = keep
    k = [i for i, ln in enumerate(lines) if ln.startswith('v_add3_u32')][0]
    return lines[:k], [ln.replace('%[chan4]', '%[pchan]') for ln in lines[k:]]


def phase2_block(tag):
    """the pending flush, by kind (pend = 1: scalar column, 2: vector column)"""
    _, s2 = flush_split('S', False)
    _, v2 = flush_split('V', False)
    return (['s_cmp_eq_u32 %[pend], 2', f's_cbranch_scc1 .L{tag}v_%='] + s2 + [f's_branch .L{tag}d_%=', f'.L{tag}v_%=:'] + v2 + [f'.L{tag}d_%=:', 's_mov_b32 %[pend], 0'])


def kind_body(kind):
    body = []
    if kind.startswith('TV'):
        body += [f'ds_read_b128 v[0:3], %[fpa] offset:16', f'ds_read_b128 v[4:7], %[fpa] offset:32']
        ms = G.main_stream(kind, packed=False)
        out, seen = [], 0
        for it in ms:
            if isinstance(it, tuple):
                if it[1] == 'f1':
                    out.append('s_waitcnt lgkmcnt(1)')
                elif it[1] == 'f2':
                    out.append('s_waitcnt lgkmcnt(0)')
                continue
            out.append(it)
        return body + fix(out)
    return fix(G.main_stream(kind, packed=False))


def epilogue(ST):
    SO, SW = ST * TB, ((ST + 3) & 3) * TB
    L = []
    ABL = os.environ.get('GEN_ABL', '')          # timing-only ablations (results invalid): nobias, noring, nofold, nofma
    if 'nobias' not in ABL:
        for j in range(4):
            L.append(f'ds_read_b128 v[{4 * j}:{4 * j + 3}], %[ringb] offset:{SO + BIAS_OFF + 16 * j}')
    # this thread's two chunks of record t+3 (requested at the start of the burst) into the stage tile t-1 has left
    RING = os.environ.get('GEN_RING', 'first')       # where the ring stores sit: first (default) | mid (behind the fold) | late (behind the tile kind's FMAs)
    ring = ['s_waitcnt vmcnt(1)', f'ds_write_b128 %[rw0], %[st0] offset:{SW}', 's_waitcnt vmcnt(0)', f'ds_write_b128 %[rw1], %[st1] offset:{SW}'] if 'noring' not in ABL else []
    if RING == 'first':
        L += ring
    if not os.environ.get('GEN_ONE_ACC') and 'nofold' not in ABL:      # (experiment: one accumulator chain, nothing to fold)
        for r in range(16):
            L.append(f'v_add_f32 v{16 + r}, v{16 + r}, v{32 + r}')
    if 'nobias' not in ABL:
        L.append('s_waitcnt lgkmcnt(2)' if ('noring' not in ABL and RING == 'first') else 's_waitcnt lgkmcnt(0)')
        for r in range(16):
            L.append(f'v_fmac_f32 v{16 + r}, v{r}, %[bsc2]')
    if RING == 'mid':
        L += ring
    # tile kind -> selector (1 RA, 2 RT, 3 + cross bits TV, 7 RTS)
    L += ['s_and_b32 %[t0], %[w0], 3', 's_bfe_u32 %[t1], %[w0], 0x2000e', 's_add_i32 %[t1], %[t1], 3', 's_add_i32 %[t2], %[t0], 1',
          's_cmp_eq_u32 %[t0], 2', 's_cselect_b32 %[t2], %[t1], %[t2]', 's_cmp_eq_u32 %[t0], 3', 's_cselect_b32 %[sel], 7, %[t2]']
    one = (lambda b: b[:1]) if 'nofma' in ABL else (lambda b: b)
    variants = [(1, one(kind_body('RA'))), (2, one(kind_body('RT')))] + [(3 + x, kind_body(f'TV{x}')) for x in range(4)] + [(7, kind_body('RTS'))]
    inl, ool_k = G.dispatch_split(variants, 'sel', 'k')
    L += inl
    if RING == 'late':
        L += ring
    # packed quad (6-channel columns: accumulator quad 3 carries another a / c row quad for channel pair xp)
    L += ['s_bitcmp0_b32 %[w0], 7', 's_cbranch_scc1 .Lnp_%=',
          's_lshr_b32 %[t0], %[w0], 6', 's_and_b32 %[t0], %[t0], 0xf0', 'v_add_u32 v52, %[t0], %[fra]', 'ds_read_b128 v[8:11], v52',
          's_bfe_u32 %[t1], %[w0], 0x20008',
          's_cmp_eq_u32 %[t1], 0', 's_cselect_b32 %[pk0], 1.0, 0', 's_cmp_eq_u32 %[t1], 1', 's_cselect_b32 %[pk1], 1.0, 0', 's_cmp_eq_u32 %[t1], 2', 's_cselect_b32 %[pk2], 1.0, 0',
          's_waitcnt lgkmcnt(0)']
    L += fix([it for it in G.packed_stream() if not isinstance(it, tuple)])
    L.append('.Lnp_%=:')
    # flush of the column this tile closes
    if 'noflushA' in ABL:      # (timing only: group A never flushes - what would aligning A's flush with B's be worth?)
        L += ['s_cmp_eq_u32 %[grp], 0', 's_cbranch_scc1 .Lnf_%=']
    if DEFER:
        # The two half-groups of a workgroup run the same tile one interval apart (k_conv_x.hip), so a flush tile's long epilogue lengthens TWO consecutive intervals:
        # group A's in the tile's own interval, group B's in the next one (profiles/r05_conv_kernel_experiments.md: 4 % of the tile loop, half of it the double counting).
        # Group A therefore keeps the column's values in P0..P8 and runs scan + atomics in its NEXT epilogue - the interval in which group B flushes the same column.
        L += ['s_cmp_eq_u32 %[pend], 0', 's_cbranch_scc1 .Lown_%=', 's_mov_b32 %[t2], 0', '.Lp2_%=:'] + phase2_block('q')
        L += ['s_cmp_eq_u32 %[t2], 1', 's_cbranch_scc1 .Lnf_%=', '.Lown_%=:']
        L += ['s_bfe_u32 %[t0], %[w0], 0x20002', 's_cmp_eq_u32 %[t0], 0', 's_cbranch_scc1 .Lnf_%=', 's_cmp_eq_u32 %[t0], 2', 's_cbranch_scc1 .Lfv_%=',
              's_cmp_eq_u32 %[sel], 7', 's_cbranch_scc1 .Lfr_%=']
        s1, r1, v1 = flush_split('S', False)[0], flush_split('S', True)[0], flush_split('V', False)[0]
        L += s1 + ['s_mov_b32 %[pend], 1', 's_branch .Lset_%=', '.Lfr_%=:'] + r1 + ['s_mov_b32 %[pend], 1', 's_branch .Lset_%=', '.Lfv_%=:'] + v1 + ['s_mov_b32 %[pend], 2']
        L += ['.Lset_%=:', 's_mov_b32 %[pchan], %[chan4]', 's_cmp_eq_u32 %[grp], 0', 's_cbranch_scc1 .Lnf_%=', 's_mov_b32 %[t2], 1', 's_branch .Lp2_%=']
    else:
        L += ['s_bfe_u32 %[t0], %[w0], 0x20002', 's_cmp_eq_u32 %[t0], 0', 's_cbranch_scc1 .Lnf_%=', 's_cmp_eq_u32 %[t0], 2', 's_cbranch_scc1 .Lfv_%=',
              's_cmp_eq_u32 %[sel], 7', 's_cbranch_scc1 .Lfr_%=']
        L += flush('S', False) + ['s_branch .Lnf_%=', '.Lfr_%=:'] + flush('S', True) + ['s_branch .Lnf_%=', '.Lfv_%=:'] + flush('V', False)
    L += ['.Lnf_%=:', 's_waitcnt lgkmcnt(0)', 's_branch .Lexit_%='] + ool_k + ['.Lexit_%=:']
    return L


def main():
    out = ['// GENERATED by tools/gen_conv_x_epi.py - do not edit.\n']
    accs = [o for o in G.acc_ops('') if not o[0].startswith('R')]
    outs = [('D0', '+{v[16:31]}', 'D0')] + accs + [(n, '=&s', f'{n}_') for n in ('t0', 't1', 't2', 'sel', 'pk0', 'pk1', 'pk2', 'sv')]
    one_acc = bool(os.environ.get('GEN_ONE_ACC'))      # one accumulator chain (the two-limb form, k_conv_x.hip X3_TWO_LIMBS -> k_conv_x2.hip): no D1 operand, nothing to fold
    ins = ([] if one_acc else [('D1', '{v[32:47]}', 'D1')]) + [('f0', '{v[48:51]}', 'f0'), ('st0', 'v', 'st0'), ('st1', 'v', 'st1'), ('ringb', 'v', 'ringb_u'), ('rw0', 'v', 'ringw0_u'),
           ('rw1', 'v', 'ringw1_u'), ('fpa', 'v', 'fpa_u'), ('fra', 'v', 'fra_u'), ('bsc2', 'v', 'bsc2'), ('oscv', 'v', 'oscv'), ('s0', 'v', 's0'), ('vx', 'v', 'vx'),
           ('vy', 'v', 'vy'), ('vz', 'v', 'vz'), ('sm1', 'v', 'seg.m1'), ('sm2', 'v', 'seg.m2'), ('sm4', 'v', 'seg.m4'), ('sm8', 'v', 'seg.m8'), ('sm16', 'v', 'seg.m16'),
           ('vrow', 'v', 'vrow'), ('hh4', 'v', 'hh4_u'), ('hh12', 'v', 'hh12_u'), ('w0', 's', 'w0'), ('chan4', 's', 'chan4_'), ('tail', 's', 'tail_mask'),
           ('sumbase', 's', 'sumbase')] + ([('grp', 's', 'grp_i')] if ('noflushA' in os.environ.get('GEN_ABL', '') and not DEFER) else [])
    if DEFER:
        pend_ops = [(f'P{i}', '+v', f'P_[{i}]') for i in range(9)] + [('pend', '+s', 'pend_'), ('pchan', '+s', 'pchan_')]
        outs = outs[:1] + accs + pend_ops + outs[1 + len(accs):]
        ins = ins + [('grp', 's', 'grp_i')]
    for ST in range(4):
        lines = epilogue(ST)
        txt = G.asm_stmt(f'X3_EPI_{ST}', lines, outs, ins, CLOBBER + ['memory', 'scc'])
        out.append(txt)
    if DEFER:      # behind a unit's last tile: group A's pending flush (group B has none)
        d_outs = [(f'P{i}', '+v', f'P_[{i}]') for i in range(9)] + [('pend', '+s', 'pend_'), ('sv', '=&s', 'sv_')]
        d_ins = [(n, 'v', e) for n, e in (('sm1', 'seg.m1'), ('sm2', 'seg.m2'), ('sm4', 'seg.m4'), ('sm8', 'seg.m8'), ('sm16', 'seg.m16'), ('vrow', 'vrow'), ('hh4', 'hh4_u'), ('hh12', 'hh12_u'))]
        d_ins += [('pchan', 's', 'pchan_'), ('tail', 's', 'tail_mask'), ('sumbase', 's', 'sumbase')]
        lines = ['s_cmp_eq_u32 %[pend], 0', 's_cbranch_scc1 .Ldr_%='] + phase2_block('r') + ['.Ldr_%=:']
        out.append(G.asm_stmt('X3_EPI_DRAIN', lines, d_outs, d_ins, ['v53', 'memory', 'scc']))
    else:
        out.append('#define X3_EPI_DRAIN() ((void)0)\n')
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'disco_diffdock_amd', 'csrc', os.environ.get('GEN_OUT', 'k_conv_x_epi_gen.inc'))
    with open(path, 'w') as f:
        f.write('\n'.join(out))
    print('wrote', os.path.normpath(path), sum(s.count('\n') for s in out), 'lines')


if __name__ == '__main__':
    main()
