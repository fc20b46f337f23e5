"""Half-phase timeline of the default conv kernel (k_conv_x.hip) from its TRACE instantiation (ddk_debug_conv_trace): workgroup 0 of one conv
layer's launch stamps s_memtime at burst start / burst end / epilogue start / epilogue end of every tile, and at four points of every unit's
prologue.  Prints the mean spans per wave (waves 0-3 = group A, 4-7 = group B; every stamp itself costs ~100 cycles).

    python tools/conv_trace.py [--layer 3] [--t 0.6]"""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex

ap = argparse.ArgumentParser()
ap.add_argument('--layer', type=int, default=3)
ap.add_argument('--t', type=float, default=0.6)
ap.add_argument('--epi', action='store_true', help='four more stamps inside every epilogue: LDS requests + ring stores | fold | tensor product | packed quad, flush, descriptor')
ap.add_argument('--samples', type=int, default=40, help='poses in the batch (5: every node row fits every L2)')
ap.add_argument('--coarse', action='store_true', help='one record per unit (no stamps inside the tile loop): undisturbed cycles per tile')
a = ap.parse_args()
dev = torch.device('cuda:0')
ctx = Context(device=0)
ctx.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
c = synthetic.make_complex(0, n_res=300)
B = a.samples
cx = Complex(ctx, c, B)
rng = np.random.default_rng(0)
pos = torch.from_numpy(np.stack([c['lig_pos'] + rng.normal(0, 3.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)).to(dev)
for _ in range(3):
    cx.score_forward(pos, a.t, a.t, a.t)
trace = torch.zeros((8, 1024, 8), dtype=torch.int32, device=dev)
ctx._check(ctx.L.ddk_debug_conv_trace(ctx.h, a.layer + (100 if a.coarse else 200 if a.epi else 0), C.c_void_p(trace.data_ptr())), 'trace')
cx.score_forward(pos, a.t, a.t, a.t)
ctx._check(ctx.L.ddk_debug_conv_trace(ctx.h, -1, None), 'trace off')
torch.cuda.synchronize()
tr = trace.cpu().numpy().astype(np.int64) & 0xffffffff
if a.coarse:
    n = int((tr[0, :, 2] != 0).sum())
    print(f'layer {a.layer}: {n} units recorded by workgroup 0 (ticks = s_memtime)')
    for w in range(8):
        x = tr[w, :n]
        d = lambda p, q: (x[:, p] - x[:, q]) & 0xffffffff
        tiles = x[:, 1]
        full = tiles == np.max(tiles)
        print(f'wave {w}: units {n} ({int(full.sum())} with all {int(np.max(tiles))} tiles)  prologue {d(7, 4)[full].mean():7.0f}  tile loop {d(0, 7)[full].mean():8.0f} = {(d(0, 7)[full] / tiles[full]).mean():6.0f} per tile  '
              f'hand-over {d(2, 0)[full].mean():5.0f}  unit {d(2, 4)[full].mean():8.0f}  | column-split units: {int((~full).sum())}, per tile {(d(0, 7)[~full] / np.maximum(tiles[~full], 1)).mean() if (~full).any() else 0:6.0f}'
              + (f'  | k_conv_y: block a prologue {d(5, 4)[full].mean():6.0f}, block b {d(6, 5)[full].mean():6.0f}, rest {d(7, 6)[full].mean():5.0f}, drain {d(3, 0)[full].mean():5.0f}' if x[0, 3] != 0 else ''))
    sys.exit(0)
n = int((tr[0, :, 3] != 0).sum())
if a.epi:
    print(f'layer {a.layer}: epilogue sub-phases (ticks incl. ~100 per stamp), tiles that are not the first of a unit; by tile class')
    for w in (0, 4):
        x = tr[w, :n]
        ok = x[:, 4] != 0
        ok[np.nonzero(x[:, 4] != 0)[0]] = True
        # a unit's first tile carries the prologue stamps in slots 4-7: recognise it by slot 7 < slot 0 (prologue precedes the burst)
        notfirst = (x[:, 4] > x[:, 2]) & (x[:, 7] > x[:, 4])
        if not notfirst.any():
            print('no sub-stamps found; first records:'); print(x[:6])
        y = x[notfirst]
        ep = y[:, 3] - y[:, 2]
        parts = np.stack([y[:, 4] - y[:, 2], y[:, 5] - y[:, 4], y[:, 6] - y[:, 5], y[:, 7] - y[:, 6], y[:, 3] - y[:, 7]], 1)
        for name, m in (('ordinary (epilogue < 1400)', ep < 1400), ('long (>= 1400: flush / vector tiles)', ep >= 1400)):
            if m.any():
                print(f'wave {w} {name}: {int(m.sum())} tiles, epilogue {ep[m].mean():.0f} = requests+stores {parts[m, 0].mean():.0f} | fold {parts[m, 1].mean():.0f} | tensor product {parts[m, 2].mean():.0f} | quad+flush+descriptor {parts[m, 3].mean():.0f} | close {parts[m, 4].mean():.0f};  burst {(y[m, 1] - y[m, 0]).mean():.0f}')
    sys.exit(0)
print(f'layer {a.layer}: {n} tiles recorded by workgroup 0 (ticks = shader cycles)')
for w in range(8):
    x = tr[w, :n]
    burst, bar1, epi = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2]
    nxt = x[1:, 0] - x[:-1, 3]
    same = x[1:, 4] == 0                      # the next tile belongs to the same unit
    first = x[:, 4] != 0
    pro = [int((x[first, b] - x[first, a_]).mean()) for a_, b in ((4, 5), (5, 6), (6, 7), (7, 0))]
    print(f'wave {w}: burst {burst.mean():6.0f}  wait {bar1.mean():5.0f}  epilogue {epi.mean():6.0f} (median {np.median(epi):5.0f}, p90 {np.percentile(epi, 90):5.0f})  '
          f'wait {nxt[same].mean():5.0f}  period {(x[1:, 0] - x[:-1, 0])[same].mean():6.0f} | prologue of {int(first.sum())} units: indices + staging {pro[0]}, '
          f'GEMM1 {pro[1]}, limbs + F rows {pro[2]}, barrier + first fragments {pro[3]}')

# per tile position inside a unit (wave 0 = group A, wave 4 = group B): which tiles carry the long epilogues (flush tiles, vector tiles)
if os.environ.get('CONV_TRACE_PER_TILE'):
    for w in (0, 4):
        x = tr[w, :n]
        starts = np.nonzero(x[:, 4] != 0)[0]
        if len(starts) < 3:
            continue
        T = int(np.median(np.diff(starts)))
        full = [s for s in starts[:-1] if (x[s + 1:s + T, 4] == 0).all() and s + T <= n]
        ep = np.stack([(x[s:s + T, 3] - x[s:s + T, 2]) for s in full]).mean(0)
        bu = np.stack([(x[s:s + T, 1] - x[s:s + T, 0]) for s in full]).mean(0)
        print(f'wave {w}: {len(full)} units of {T} tiles; epilogue by tile position:')
        print('   ' + ' '.join('%4d' % v for v in ep))
        print('   burst by tile position:')
        print('   ' + ' '.join('%4d' % v for v in bu))
        pe = np.stack([(x[s + 1:s + T, 0] - x[s:s + T - 1, 0]) for s in full]).mean(0)
        print('   period (burst start to next burst start) by tile position; sum over the unit %d, %d tiles x the median period = %d:' % (pe.sum(), T - 1, (T - 1) * np.median(pe)))
        print('   ' + ' '.join('%4d' % v for v in pe))
