#!/usr/bin/env python
"""bench.py - headline benchmark of the ddk hot path (BASELINE.json metric):

    complexes/sec, 20-step 40-sample reverse-diffusion inference

``value`` uses the REFERENCE'S OWN BRACKET (evaluate.py:259,293): the wall time around ``sampling(data_list, model, ...)`` on host
``data_list``s of 40 graph copies with host start poses, a NEW complex every call - collation, ``ddk_complex_create`` (static
precompute + H2D), the pose upload, the device noise draws, the 20-step loop and the pose write-back are all inside; K calls
back to back, one device synchronisation at the end, value = K / wall (x world).  One "step" = one complex.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

configs (BASELINE.json ``configs``; the driver's N = 1 line is config 2):
  2  DiffDock-S score model, 8 synthetic complexes (~30 ligand atoms / 300 C-alpha) per GPU, 40 samples, 20 steps
  3  DisCo-DiffDock-S score model (latent_dim = 2) + AR latent model
  4  config 3 + all-atom confidence model on the final poses, confidences gathered with the poses
  5  large-pocket stress: 2000 C-alpha; the 40 samples of every complex are sharded over the ranks (N = 1: all 40 on one GPU)
N > 1: one process per GPU (torch.distributed, backend nccl == RCCL); configs 2-4 shard COMPLEXES (weak scaling: K complexes per
rank), config 5 shards SAMPLES (strong scaling); no collective on the data path, one final all_gather of the poses.

Without torchrun, ``--gpus N`` (N > 1) starts the N ranks itself (re-exec under ``python -m torch.distributed.run --nproc-per-node N``).

Prints ONE JSON line on rank 0, with
  roofline      fused TP-conv kernel, HIP events around every launch of the timed region: ``achieved`` / ``frac`` = MFMA FLOPs the launches
                EXECUTED / event time vs the dense f16 matrix peak (the kernel's own roofline fraction); ``fp32_equivalent_TFLOPs`` = the
                algorithmic fp32 FLOPs of the evaluated edges over the same time; ``reference_equivalent_TFLOPs`` also counts the rec-rec
                messages the receptive-field pruning proves dead (no frac)
  extra         pruning_off: the same bracket with ddk_set_receptive_field_pruning(ctx, 0) - the guaranteed floor of ``value``;
                pocket_bound: the same bracket on start poses inside the pocket with noise scaled so that every sample keeps its
                cross edges for all 20 steps (what a trained model's trajectories look like to the pruning);
                per_step: executed-edge fraction and conv ms of each of the 20 reverse steps of the default workload;
                create_ms: host time of one ddk_complex_create call at 300 / 2000 residues; per_call_spread_same_complex: max / min - 1 of
                the device time of the timed calls that ran the same complex;
                device_loop: the same workload with complexes and noise resident in HBM (round 1's bracket), for comparison
  fallback_fp32_kernel  the resident loop with ddk_config.conv_kernel = 1 (fp32 MFMA chains), N = 1 and config 2 only
  cpu_baseline  the CPU oracle (oracle/: PyTorch-CPU restatement of the reference) on this box's host cores, bounded sample;
                its scores are asserted equal to the GPU's on the same inputs
"""
import argparse
import copy
import json
import os
import sys
import tempfile
import time
from argparse import Namespace
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLES, STEPS, N_COMPLEXES = 40, 20, 8
README_S = dict(temp_sampling=[1.886430780895051, 5.659562317960644, 2.8888668488630156],
                temp_psi=[0.07085125444659945, 2.686505606141324, 4.089493860493927],
                temp_sigma_data=[0.3617563913086843, 0.7437588205919711, 0.08897393057297842])
README_DISCO = dict(temp_sampling=[1.546842681537956, 4.005218254154881, 3.6499018519649384],
                    temp_psi=[1.2685697872473618, 1.2760150490206228, 2.0625243924678136],
                    temp_sigma_data=[0.8456140350087653, 0.453446580767075, 0.3292199987743284])
W_LAYER = [720, 936, 1152, 1872, 1872]
TP_FLOP = [2016, 2736, 3456, 5472, 5472]
FUSED_BYTES = [408, 480, 552, 648, 648]                        # fused boundary, bytes per edge
PEAK_F32_MFMA_TFLOPS = 157.3                                   # MI355X_MICROARCH.md
PEAK_F16_MFMA_TFLOPS = 2500.0                                  # dense f16 / bf16 matrix peak (MI355X_MICROARCH.md; micro-benchmark ceiling 2382)
MFMA_FLOP_PER_EDGE_TILE = {6: (24 * 2 * 32 * 32 * 16 + 6 * 2 * 32 * 32 * 8) / 32,      # k_conv_x.hip (conv_kernel = 3, six limb products): 432 K-columns of 32x32 per 32 edges and W2 tile (26 x 32x32x16 + 2 x 32x32x8 with the packed tail; 24 + 6 unpacked; the same count per GEMM1)
                           3: (14 * 2 * 32 * 32 * 16) / 32}      # k_conv_x2.hip (the default: two limbs, three limb products): 224 K-columns = 14 x 32x32x16 per tile (12 + the packed K = 8 tail's two, which carry mid.mid for free)
PEAK_HBM_GBS = 8000.0

ARGS_S = Namespace(ns=24, nv=6, num_conv_layers=5, sigma_embed_dim=32, distance_embed_dim=32, cross_distance_embed_dim=32,
                   max_radius=5.0, cross_max_distance=80, dynamic_max_cross=True, embedding_scale=1000, embedding_type='sinusoidal',
                   scale_by_sigma=True, no_torsion=False, no_batch_norm=False, dropout=0.1, sh_lmax=1, use_second_order_repr=False,
                   use_old_atom_encoder=False, esm_embeddings_path='data/esm2_3billion_embeddings.pt', latent_dim=0, latent_vocab=64,
                   latent_cross_attention=False, tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55,
                   tor_sigma_min=0.03, tor_sigma_max=3.14)
ARGS_DISCO = Namespace(**dict(vars(ARGS_S), latent_dim=2, latent_vocab=1, latent_droprate=0.1))
ARGS_AR = Namespace(use_pretrained_score=True, ns=16, latent_no_batchnorm=False, latent_dropout=0.0, latent_hidden_dim=128,
                    esm_embeddings_path='x', no_randomness=False)
ARGS_CONF = Namespace(all_atoms=True, ns=24, nv=6, num_conv_layers=5, sigma_embed_dim=32, distance_embed_dim=32, cross_distance_embed_dim=32,
                      max_radius=5.0, cross_max_distance=80, dynamic_max_cross=True, embedding_type='sinusoidal', embedding_scale=10000,
                      scale_by_sigma=True, no_torsion=False, no_batch_norm=False, dropout=0.1, use_second_order_repr=False,
                      esm_embeddings_path='data/esm2_3billion_embeddings.pt', rmsd_classification_cutoff=[2.0], confidence_no_batchnorm=False,
                      tr_sigma_min=0.1, tr_sigma_max=34.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.0314, tor_sigma_max=3.14)

CONFIG_TEXT = {
    2: 'DiffDock-S score model (random-init weights, reference state_dict layout), 8 synthetic complexes per GPU (20-40 ligand atoms / 300 '
       'C-alpha, 24-NN receptor graph), samples_per_complex=40, inference_steps=20, README low-temperature sampling, no_final_step_noise',
    3: 'DisCo-DiffDock-S score model (latent_dim=2, latent_vocab=1) + AR latent model (PretrainedScoreEncoder, softmax temperature e^-1.5), '
       '8 synthetic complexes per GPU (300 C-alpha), samples_per_complex=40, inference_steps=20, README DisCo temperatures',
    4: 'config 3 + all-atom confidence model (paper_confidence_model layout, ~2400 receptor atoms) on the final poses; confidences gathered '
       'with the poses',
    5: 'large-pocket stress: DiffDock-S score model, synthetic complexes with 2000 C-alpha (24-NN within 15 A), samples_per_complex=40 '
       'sharded over the ranks, inference_steps=20',
}


CONV_KERNEL_SOURCES = ('k_conv_x.hip', 'k_conv_x2.hip', 'k_conv_x_epi_gen.inc', 'k_conv_x_epi2_gen.inc', 'k_conv_common.h', 'ddk_internal.h')      # what the dominant kernel is compiled from


def conv_kernel_source_sha():
    """sha256 over the sources of the dominant kernel: a PMC profile is only quoted for the kernel it was taken on"""
    import hashlib
    h = hashlib.sha256()
    for name in CONV_KERNEL_SOURCES:
        with open(os.path.join(ROOT, 'disco_diffdock_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic():
    """HBM bytes per fused-conv launch from the committed PMC passes of this same command - only when the profile was taken on THIS kernel:
    tools/summarize_profile.py stamps the profile with the sha256 of the kernel's sources, a profile of another (or no) hash yields None and
    the reason."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True)
    if not cands:
        return None, 'no profiles/r*_pmc_traffic.json'
    name = os.path.basename(cands[0])
    try:
        d = json.load(open(cands[0]))
        if d.get('kernel_source_sha256') != conv_kernel_source_sha():
            return None, (f'profiles/{name} was taken on other kernel sources (profile {str(d.get("kernel_source_sha256"))[:12]}, working tree '
                          f'{conv_kernel_source_sha()[:12]}): re-run tools/profile_round.sh')
        return d['traffic_bytes_per_launch'], name
    except Exception as e:
        return None, f'profiles/{name}: {e}'


def start_poses(c, rng, samples, ctx=None, dev=None, tr_sigma_max=19.0):
    """randomize_position (reference utils/sampling.py:12-34) on the device (ddk_randomize_position): uniform(-pi, pi) torsion angles on every
    rotatable bond (:16-22), a random rotation about the centroid and a N(0, tr_sigma_max) translation (:24-34) per sample.
    ctx = None (the CPU-oracle sample, development tools): host arithmetic, rigid part only."""
    from scipy.spatial.transform import Rotation
    if ctx is None:
        lp = c['lig_pos'].astype(np.float64)
        ctr = lp.mean(0, keepdims=True)
        return np.stack([(lp - ctr) @ Rotation.random(random_state=rng).as_matrix().T + rng.normal(0, tr_sigma_max, size=(1, 3))
                         for _ in range(samples)]).astype(np.float32)
    from disco_diffdock_amd.runtime import Complex
    cx = Complex(ctx, {k: v for k, v in c.items() if not k.startswith('atom_')}, samples)
    rot = np.stack([Rotation.random(random_state=rng).as_matrix() for _ in range(samples)]).astype(np.float32)
    tor = rng.uniform(-np.pi, np.pi, size=(samples, cx.R)).astype(np.float32) if cx.R > 0 else None
    tr = rng.normal(0, tr_sigma_max, size=(samples, 3)).astype(np.float32)
    out = cx.randomize_position(torch.from_numpy(np.array(c['lig_pos'], np.float32)).to(dev), torch.from_numpy(rot).to(dev),
                                None if tor is None else torch.from_numpy(tor).to(dev), torch.from_numpy(tr).to(dev)).cpu().numpy()
    cx.close()
    return out.astype(np.float32)


def cpu_baseline(c, P, coeffs, gpu_scores, n_res, poses, seconds_budget=150.0, chunk=8):
    """The CPU oracle on a bounded sample of the SAME workload: the full batch (40 samples, the bench's own start poses of this complex: randomize_position
    with tr_sigma_max = 19) x ONE reverse step (score model forward + modify_conformer_batch) at t in {1.0, 0.5, 0.05} (the cross graph shrinks with t), evaluated in
    chunks of `chunk` samples (the samples of a batch are independent; the reference's own fallback halves the batch the same way, evaluate.py:397-398)
    because the materialised [E, W] weights of 40 samples do not fit comfortably.  No extrapolation over the batch; x 20 steps over the trajectory.  The oracle's
    scores are compared with the GPU's on the same inputs (the oracle is the checker here, never the path)."""
    from oracle import score_model_ref as smr, sampler_ref as spr, graph_lite
    cfg = smr.ScoreModelConfig(latent_vocab=64)
    d = os.path.join(ROOT, 'disco_diffdock_amd', 'data')
    tables = (np.load(os.path.join(d, 'so3_exp_score_norms.npy')), np.load(os.path.join(d, 'torus_score_norm_seed0.npy')))
    b = len(poses)

    def graph():
        g = graph_lite.make_complex(c['lig_x'], c['lig_pos'], c['bond_index'], c['bond_attr'], c['edge_mask'], c['mask_rotate'],
                                    c['rec_x'], c['rec_pos'], c['rec_edge_index'])
        g['ligand'].mask_rotate = [g['ligand'].mask_rotate]
        return g
    t_arr, sc, nc = coeffs

    def step(pos, t):
        dl = [graph() for _ in range(len(pos))]
        for g, p in zip(dl, pos):
            g['ligand'].pos = torch.from_numpy(p)
        batch = graph_lite.collate(dl)
        t0 = time.perf_counter()
        with torch.no_grad():
            spr.set_time(batch, t, t, t, len(pos))
            tr, rot, tor = smr.score_model_forward(P, cfg, batch, tables[0], tables[1])
            spr.modify_conformer_batch(batch['ligand'].pos, batch, sc[0, 0] * tr, sc[0, 1] * rot, sc[0, 2] * tor, torch.from_numpy(np.asarray(c['mask_rotate'])))
        return time.perf_counter() - t0, tr, rot, tor
    step(poses[:2], 0.5)                 # warm-up (thread pool, lazy initialisation)
    per_t, worst, t_start = {}, 0.0, time.perf_counter()
    for t in (1.0, 0.5, 0.05):
        if per_t and time.perf_counter() - t_start > seconds_budget * len(per_t) / 3.0 + seconds_budget / 3.0:
            break                        # (a slow host: fewer diffusion times, said in `sample`)
        tot, outs = 0.0, []
        for a0 in range(0, b, chunk):
            dt, tr, rot, tor = step(poses[a0:a0 + chunk], t)
            tot += dt
            outs.append((tr, rot, tor))
        per_t[t] = tot
        tr, rot, tor = (torch.cat([o[k] for o in outs]) for k in range(3))
        g_tr, g_rot, g_tor = gpu_scores(poses, t)
        for a_, r in ((g_tr, tr), (g_rot, rot), (g_tor, tor)):
            if r.numel():
                worst = max(worst, float((a_.cpu() - r).abs().max() / r.abs().max()))
    assert worst < 1e-4, f'GPU scores differ from the CPU oracle on the cpu_baseline sample: {worst:.2e}'
    t_step = float(np.mean(list(per_t.values())))
    per_complex = t_step * STEPS
    return dict(value=1.0 / per_complex, unit='complexes/s', cores=torch.get_num_threads(), kind='port',
                gpu_vs_oracle_rel_err=worst,
                sample=f'oracle (PyTorch-CPU restatement, materialised [E,W] weights): {b} samples (the full batch, the bench\'s own start poses, in chunks of {chunk}) x 1 '
                       f'reverse step of one {n_res}-residue complex at t = ' + ' / '.join(f'{t:g}' for t in per_t) + ': ' + ' / '.join(f'{per_t[t]:.1f}' for t in per_t) +
                       f' s, mean {t_step:.1f} s per step x{STEPS} steps (no extrapolation over the batch); oracle scores == GPU scores to {worst:.1e}')


def pocket_poses(c, rng, samples, sigma=1.0):
    """start poses of the pocket-bound workload: the ligand at its pocket position, random rotation about the centroid, N(0, sigma) A
    translation - every sample starts (and, with the scaled noise, stays) in cross-edge contact with the receptor"""
    from scipy.spatial.transform import Rotation
    lp = c['lig_pos'].astype(np.float64)
    ctr = lp.mean(0, keepdims=True)
    return np.stack([(lp - ctr) @ Rotation.random(random_state=rng).as_matrix().T + ctr + rng.normal(0, sigma, size=(1, 3))
                     for _ in range(samples)]).astype(np.float32)


def _make_complex(job):
    """one synthetic complex of the workload (top level: the large sets are generated by a process pool)"""
    from disco_diffdock_amd import synthetic
    seed, n_res, n_lig, with_conf = job
    c = synthetic.make_complex(seed, n_res=n_res, n_lig=n_lig)
    if with_conf:
        synthetic.add_receptor_atoms(c, np.random.default_rng(seed))
    return c


def set_shapes(n_cx, offset, n_res_default, spread, fixed_receptor):
    """(n_res, n_lig or None) of the n_cx complexes of this run: the default 8 share one receptor size and the generator's own ligand; a larger set is
    timesplit-SHAPED (synthetic.timesplit_shape: log-normal receptor sizes, 10-80-atom ligands) unless --fixed-receptor keeps round 5's 300-residue stream"""
    from disco_diffdock_amd import synthetic
    if not spread:
        return [(n_res_default, None) for _ in range(n_cx)]
    sh = [synthetic.timesplit_shape(offset + i) for i in range(n_cx)]
    return [(n_res_default if fixed_receptor else r, l) for r, l in sh]


def tp_boundary_a(dev, edges=800000, iters=10):
    """BASELINE metric, second clause, at the REFERENCE's op boundary: FasterTensorProduct.forward (models/tensor_layers.py:65-116) with the per-edge weights
    [E, W] resident in HBM - 4 (W + din + 4 + dout) + 8 B per edge for ~5 kFLOP: the one HBM-bound kernel of the path (k_tp.hip, ddk_tp_forward).  HIP events
    on the launch stream around `iters` launches per conv-layer shape; layer 3 (= 4) is the BASELINE shape.  tools/bench_tp.py is the same measurement with
    an fp64 check of a slice; tools/profile_tp.sh takes the rocprofv3 kernel trace and the FETCH / WRITE counters of it (profiles/r06_tp_boundary_kernel_stats.md)."""
    from disco_diffdock_amd.tensor_layers import FasterTensorProduct
    seq = ['24x0e', '24x0e+6x1o', '24x0e+6x1o+6x1e', '24x0e+6x1o+6x1e+24x0o']
    din = [24, 42, 60, 84, 84]
    layers = {}
    g = torch.Generator(device=dev).manual_seed(0)
    for l in (3, 2, 1, 0):
        tp = FasterTensorProduct(seq[min(l, 3)], '1x0e+1x1o', seq[min(l + 1, 3)])
        W = tp.weight_numel
        x = torch.randn(edges, din[l], device=dev, generator=g)
        sh = torch.randn(edges, 4, device=dev, generator=g)
        w = torch.randn(edges, W, device=dev, generator=g)
        out = tp(x, sh, w)
        tp(x, sh, w)
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        st.record()
        for _ in range(iters):
            tp(x, sh, w)
        en.record()
        torch.cuda.synchronize()
        ms = st.elapsed_time(en) / iters
        b = 4 * (W + din[l] + 4) + 8 + 4 * out.shape[1]
        layers[l] = {'layer': l, 'W': W, 'algorithmic_bytes_per_edge': b, 'ms_per_launch': round(ms, 4), 'GBps': round(edges * b / ms / 1e6, 1)}
        del x, sh, w, out
    torch.cuda.empty_cache()
    l3 = layers[3]
    return {'kernel': 'ddk::tp_col_kernel (k_tp.hip): FasterTensorProduct.forward, weights [E, W] streamed from HBM', 'bound': 'hbm', 'edges': edges,
            'achieved': l3['GBps'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(l3['GBps'] / PEAK_HBM_GBS, 4),
            'frac_of_achievable_6300': round(l3['GBps'] / 6300.0, 4), 'layer': 3, 'per_layer': [layers[l] for l in (0, 1, 2, 3)],
            'note': 'layer 3 = layer 4 = the BASELINE shape (W = 1872, 8 184 B per edge); 6 300 GB/s is what MI355X_MICROARCH.md measures as achievable (float4 copy)'}


def timesplit_stream():
    """north_star's own workload inside the driver's line: BASELINE config 4 (DisCo-DiffDock-S + AR latent model + all-atom confidence model)
    over 363 DISTINCT synthetic complexes with a 10-80-atom ligand spread (a timesplit_test-sized set, reference README.md:20, evaluate.py:221-293), every
    complex streamed ONCE through sampling() - as its own process behind this one's measurements (the GPU is idle by then), the three brackets of the headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--config', '4', '--complexes', '363', '--steps', '363', '--warmup', '2', '--no-cpu-baseline', '--no-alt',
           '--no-device-loop', '--no-timesplit']
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not line:
            return {'error': f'rc {r.returncode}: ' + (r.stderr or '')[-400:]}
        o = json.loads(line[-1])
    except Exception as e:      # noqa: BLE001  (the headline must not die with its extra)
        return {'error': repr(e)}
    return {'command': ' '.join(cmd[1:]), 'workload': o['config']['workload'], 'value': o['value'], 'value_pruning_off': o.get('value_pruning_off'),
            'value_pocket_bound': o.get('value_pocket_bound'), 'unit': 'complexes/s', 'complexes': 363, 'ms_per_complex': o['ms_per_step'],
            'receptor_residues_min_median_max': o['extra']['stream'].get('receptor_residues_min_median_max'), 'per_receptor_size_decile': o['extra'].get('per_receptor_size_decile'),
            'roofline_frac': o['roofline']['frac'], 'avg_conv_launch_ms': o['roofline']['avg_launch_ms'], 'conv_share_of_wall': o['roofline']['conv_share_of_wall'], 'conv_share_of_wall_incl_ar_model': o['roofline'].get('conv_share_of_wall_incl_ar_model'),
            'min_cross_edges_per_sample_pocket_bound': (o['extra'].get('pocket_bound') or {}).get('min_cross_edges_per_sample_over_steps'),
            'stream': o['extra']['stream'], 'wall_s_of_the_subprocess': round(time.perf_counter() - t0, 1)}


def self_launch(a):
    """``python bench.py --gpus N`` without torchrun: start the N ranks (one per GPU) ourselves and hand over to them"""
    import socket
    import subprocess
    if not a.single_device and torch.cuda.device_count() < a.gpus:
        sys.exit(f'bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} GPU(s) are visible (use --single-device for a smoke test of the '
                 'N > 1 path on one GPU)')
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the brackets of the other conv kernels: the three-limb / six-product form (value_six_limb_products) and the fp32-MFMA fallback')
    ap.add_argument('--no-device-loop', action='store_true', help='skip the resident-loop comparison figure')
    ap.add_argument('--no-extras', action='store_true', help='skip the pruning-off and pocket-bound brackets')
    ap.add_argument('--shard-set', action='store_true', help='configs 2-4: the --complexes complexes are ONE set, partitioned over the ranks with distributed.shard_indices (LPT by '
                    'n_rec * n_lig) and streamed once; RNG seeded per complex, so that the gathered poses do not depend on the number of ranks (extra.pose_digest)')
    ap.add_argument('--dump-poses', default=None, help='rank 0 writes the gathered final poses (npz, one array per complex) here: the N = 1 / N = 8 comparisons of tools/ranks8_check.sh')
    ap.add_argument('--no-timesplit', action='store_true', help='skip extra.timesplit_stream (config 4 over 363 distinct complexes, its own process)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl == RCCL; gloo for smoke tests)')
    ap.add_argument('--single-device', action='store_true', help='smoke test of the N > 1 path on a one-GPU box: every rank uses cuda:0')
    ap.add_argument('--complexes', type=int, default=0, help='distinct synthetic complexes per rank (default 8 of ~30 ligand atoms; N > 8: ligand sizes '
                    'drawn from 10-80 atoms like a PDBBind split - e.g. --complexes 363 --steps 363 streams a timesplit_test-sized set once)')
    ap.add_argument('--fixed-receptor', action='store_true', help='--complexes > 8: keep every receptor at 300 residues (rounds 4 / 5 streamed this set; the default draws '
                    'timesplit-shaped receptor sizes, synthetic.timesplit_shape)')
    ap.add_argument('--passes', type=int, default=0, help='timed passes over the K calls, the MEDIAN pass is reported (default: 3 for K <= 64, else 1)')
    ap.add_argument('--no-tp-boundary', action='store_true', help='skip roofline.tp_boundary_A (FasterTensorProduct.forward at the reference op boundary, 800 000 edges)')
    ap.add_argument('--complex-offset', type=int, default=0, help='first content seed of --complexes (to run a slice of a large set)')
    ap.add_argument('--force-dist', action='store_true', help='N = 1 through the N > 1 code path: a world_size-1 process group on --backend, every '
                    'barrier / all_reduce / all_gather of the path executes (proves that RCCL loads and runs beside libddk.so on a one-GPU box)')
    a = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        self_launch(a)
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        sys.exit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE = {world}')
    import torch.distributed as dist
    if a.single_device:
        local = 0
    elif torch.cuda.device_count() <= local:
        sys.exit(f'bench.py: rank {rank} has no GPU {local} ({torch.cuda.device_count()} visible)')
    use_dist = world > 1 or a.force_dist
    if use_dist:
        torch.cuda.set_device(local)
        if world > 1:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
        else:
            import socket
            with socket.socket() as so:
                so.bind(('127.0.0.1', 0))
                port = so.getsockname()[1]
            dist.init_process_group(a.backend, init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
            import disco_diffdock_amd.distributed as ddist
            ddist.FORCE_COLLECTIVES = True
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    from disco_diffdock_amd import build, synthetic, graph_cache, score_model as sm_mod
    from disco_diffdock_amd.runtime import Complex
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd.model_utils import get_model, get_ar_model
    from disco_diffdock_amd.sampling import sampling, step_coefficients, draw_noise
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from disco_diffdock_amd.distributed import shard_samples, shard_indices, complex_cost, gather_poses, gather_samples, gather_confidences
    if rank == 0:
        build.build(verbose=False)       # no-op when the shipped libddk.so is current; never build concurrently
    if use_dist:
        dist.barrier()

    cfg_id = a.config
    disco, with_conf, big = cfg_id in (3, 4), cfg_id == 4, cfg_id == 5
    n_res = 2000 if big else 300
    margs = ARGS_DISCO if disco else ARGS_S
    temps = README_DISCO if disco else README_S

    # ---- workload: this rank's complexes, written to and read back from the flat graph cache (the format a GPU box receives real
    #      PDBBind graphs in, disco_diffdock_amd/graph_cache.py) ---------------------------------------------------------------
    n_cx = a.complexes if a.complexes > 0 else (4 if big else N_COMPLEXES)
    spread_ligands = a.complexes > N_COMPLEXES
    shard_set = a.shard_set and not big
    shapes = set_shapes(n_cx, a.complex_offset, n_res, spread_ligands, a.fixed_receptor)
    if big:                                  # samples sharded: every rank works on the same complexes
        mine = list(range(n_cx))
        lo, hi = shard_samples(SAMPLES, rank, world)
    elif shard_set:                          # ONE set of n_cx complexes, partitioned by cost (the layout a real dataset gets)
        n_total = n_cx
        mine = shard_indices([complex_cost(r_, l_ or 30) for r_, l_ in shapes], rank, world)
        lo, hi = 0, SAMPLES
    else:
        n_total = n_cx * world
        # weak scaling with the per-GPU work held EXACTLY fixed: every rank holds the same n_cx receptor / ligand pairs (content seed = id mod n_cx)
        # under its own global ids, with its own start poses and noise (a real dataset is partitioned with distributed.shard_indices)
        mine = [rank * n_cx + i for i in range(n_cx)]
        lo, hi = 0, SAMPLES
    b_local = hi - lo
    if shard_set:       # this rank's timed sampling() calls: its share of the set, every complex once
        a.steps, a.warmup = len(mine), (min(a.warmup, len(mine)))
    jobs = [(a.complex_offset + i % n_cx, shapes[i % n_cx][0], shapes[i % n_cx][1], with_conf) for i in mine]
    if len(jobs) > 32:       # a timesplit-sized set: the synthetic generator (rejection sampling, ~0.25 s per complex) on the host's cores, not in a loop
        import multiprocessing as mp
        with mp.get_context('spawn').Pool(min(32, os.cpu_count() or 1)) as pool:
            made = pool.map(_make_complex, jobs, chunksize=4)
    else:
        made = [_make_complex(j) for j in jobs]
    cache_path = os.path.join(tempfile.gettempdir(), f'ddk_bench_cfg{cfg_id}_rank{rank}.ddkg')
    graph_cache.save_complexes(cache_path, made)
    complexes = dict(zip(mine, graph_cache.load_complexes(cache_path)))

    # ---- models through the reference's call surface ----------------------------------------------------------------------------
    tsig = partial(t_to_sigma, args=margs)
    P = synthetic.random_score_model_state_dict(seed=0, latent_dim=margs.latent_dim, latent_droprate=getattr(margs, 'latent_droprate', 0.0))

    def build_models(conv_kernel=None):
        """score model (+ AR latent model, + confidence model) through the reference's call surface; conv_kernel: ddk_config.conv_kernel of every context
        they create (None: the default, or whatever DDK_CONV_KERNEL says)"""
        saved = os.environ.get('DDK_CONV_KERNEL')
        if conv_kernel is not None:
            os.environ['DDK_CONV_KERNEL'] = str(conv_kernel)
        try:
            model_ = get_model(margs, dev, tsig, no_parallel=True)
            sm_ = getattr(model_, 'score_model', model_)
            sm_.load_state_dict(P, strict=True)
            extra_ = {}
            if disco:
                ar = get_ar_model(ARGS_AR, margs, dev, training=False)
                ar.load_state_dict(synthetic.random_ar_state_dict(seed=14))
                ar.eval()
                extra_.update(ar_model=ar, ar_args=ARGS_AR, softmax_latent_temperature=float(np.exp(-1.5)))
            else:
                extra_.update(use_latent=False)
            if with_conf:
                cm = get_model(ARGS_CONF, dev, partial(t_to_sigma, args=ARGS_CONF), no_parallel=True, confidence_mode=True)
                cm.load_state_dict(synthetic.random_confidence_state_dict(seed=1), strict=True)
                cm.eval()
                extra_.update(confidence_model=cm, confidence_model_args=ARGS_CONF)
        finally:
            if conv_kernel is not None:
                if saved is None:
                    os.environ.pop('DDK_CONV_KERNEL', None)
                else:
                    os.environ['DDK_CONV_KERNEL'] = saved
        return model_, sm_, extra_

    model, score_model, extra = build_models()
    models_default = (model, score_model, extra)
    sched = get_t_schedule(STEPS)
    coeffs = step_coefficients(STEPS, sched, sched, sched, tsig, margs, False, False, True, temps['temp_sampling'], temps['temp_psi'],
                               temps['temp_sigma_data'])
    t_arr, sc, nc = coeffs
    order = (mine[:a.warmup] + mine) if shard_set else [mine[k % len(mine)] for k in range(a.warmup + a.steps)]
    # rank r computes on cuda:LOCAL_RANK - the r-th device AFTER any HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES remapping - and its ddk_ctx lives there
    assert int(score_model.ctx.device) == local and torch.cuda.current_device() == local
    ctx = score_model.ctx

    # host data_lists of every call, prepared BEFORE the clock starts like evaluate.py:232-233 (deepcopy + randomize_position)
    poses_all = {i: start_poses(complexes[i], np.random.default_rng(i), SAMPLES, ctx, dev) for i in mine}
    graphs0 = {}
    for i in mine:
        c = complexes[i]
        score_only = {k: v for k, v in c.items() if not k.startswith('atom_')}
        graphs0[i] = (from_arrays(score_only), from_arrays(c) if with_conf else None)

    def data_lists(i, poses):
        g0, gc = graphs0[i]
        dl = [copy.copy(g0) for _ in range(b_local)]
        for d, p in zip(dl, poses[i][lo:hi]):
            d['ligand'].pos = torch.from_numpy(p)
            if disco:
                d['ligand'].ar_pos = torch.from_numpy(p).clone()
        cdl = [copy.copy(gc) for _ in range(b_local)] if with_conf else None
        return dl, cdl

    n_passes_default = a.passes if a.passes > 0 else (3 if a.steps <= 64 else 1)

    def bracket(poses, warmup, prune=True, noise_scale=None, n_passes=None, mdl=None):
        """the reference's bracket (evaluate.py:259,293) over K = a.steps sampling() calls, a NEW complex every call; returns the wall
        time and what the HIP events around the conv launches saw.  noise_scale: N(0,1) draws of every call pre-drawn and scaled (the
        pocket-bound workload); None: drawn inside sampling() from the device generator like the headline.  mdl: (model, score_model, extra) of build_models()
        when the bracket is to run another set of contexts (the three-limb / six-product conv kernel)."""
        model, score_model_, extra = mdl if mdl is not None else models_default
        ctx = score_model_.ctx
        n_passes = n_passes or n_passes_default
        # sampling() writes the final poses into the graphs it was given (utils/sampling.py:197-199 does the same): every pass gets its OWN host data_lists,
        # built before the clock starts, or the second pass would start from the first one's final poses
        calls_of_pass = [[data_lists(i, poses) for i in order[:warmup + a.steps]] for _ in range(n_passes)]
        calls = calls_of_pass[0]
        torch.cuda.manual_seed(977 + rank)
        noises = None
        Rs = {i: int(np.asarray(complexes[i]['mask_rotate']).shape[0]) for i in mine}
        if noise_scale is not None:
            noises = [[noise_scale * draw_noise(STEPS, b_local, Rs[i], Rs[i], nc, dev)] for i in order[:warmup + a.steps]]
        elif big and a.dump_poses:      # (comparison runs) the N(0,1) draws of ALL samples of a complex from a per-complex seed; this rank takes its samples' slice
            noises = []
            for i in order[:warmup + a.steps]:
                torch.cuda.manual_seed(5000 + i)
                noises.append([draw_noise(STEPS, SAMPLES, Rs[i], Rs[i], nc, dev)[:, lo:hi].contiguous()])
        ctx.set_pruning(prune)

        def one_call(k):
            dl, cdl = calls[k]
            kw = dict(extra)
            if with_conf:
                kw['confidence_data_list'] = cdl
            if noises is not None:
                kw['noise'] = noises[k]
            return sampling(dl, model, STEPS, sched, sched, sched, dev, tsig, margs, batch_size=b_local, no_final_step_noise=True, **temps, **kw)

        for k in range(warmup):
            one_call(k)
        torch.cuda.synchronize()
        ctx.profile_enable(True)
        ar_ctx = extra['ar_model'].pretrained_score_model.ctx if disco else None      # the AR latent model's own score-model copy: its conv launches are timed too
        if ar_ctx is not None:
            ar_ctx.profile_enable(True)
        pool0, mem_peak = ctx.pool_stats(), 0
        pass_elapsed, pass_calls = [], []
        for pass_id in range(n_passes):
            calls = calls_of_pass[pass_id]
            # every pass: EXACTLY a.steps calls between a barrier + synchronize on both sides; the same complexes, start poses and generator seed
            torch.cuda.manual_seed(4321 + rank)      # the device generator sampling() draws its noise from (the resident-loop figure below replays it)
            sm_mod._complex_cache.clear()            # a NEW complex every timed call: no Complex of the warm-up (or of the previous pass) survives
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            final, confs, outs = {}, {}, []
            call_ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]      # device time stamps behind every call (no host wait)
            call_ev[0].record()
            for k in range(warmup, warmup + a.steps):
                if (k - warmup) % len(mine) == 0 and k > warmup:
                    sm_mod._complex_cache.clear()    # K > #complexes: the second pass over the shard must not hit the cache either
                if shard_set:
                    torch.cuda.manual_seed(5000 + order[k])      # noise and AR picks of a complex do not depend on which rank runs it, or after what
                out, conf = one_call(k)
                if os.environ.get('DDK_BENCH_TRACE'):
                    print(f'[bench] call {k - warmup} complex {order[k]} n_lig {complexes[order[k]]["lig_pos"].shape[0]} queued', file=sys.stderr, flush=True)
                if os.environ.get('DDK_BENCH_SYNC'):      # debugging aid: a GPU fault then names the call it belongs to (the timing is meaningless)
                    torch.cuda.synchronize()
                    print(f'[bench] call {k - warmup} complex {order[k]} n_lig {complexes[order[k]]["lig_pos"].shape[0]} ok', file=sys.stderr, flush=True)
                call_ev[k - warmup + 1].record()
                if (k - warmup) % 8 == 0:                # device memory in use (driver query, no synchronisation), sampled every 8th call
                    fr, tot_mem = torch.cuda.mem_get_info(dev)
                    mem_peak = max(mem_peak, tot_mem - fr)
                outs.append(out)
                final[order[k]] = torch.stack([d['ligand'].pos for d in out])
                if with_conf:
                    confs[order[k]] = conf
            if disco:        # the latent bookkeeping of utils/sampling.py:205-221 (filled on first access): inside the bracket like the reference's
                assert all(len(o[0].latent_str) >= 2 and all(hasattr(d, 'latent_pos') for d in o) for o in outs)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            el_ = time.perf_counter() - t0
            if use_dist:
                tmax = torch.tensor([el_], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el_ = float(tmax.item())
            pass_elapsed.append(el_)
            pass_calls.append([round(call_ev[k].elapsed_time(call_ev[k + 1]), 2) for k in range(a.steps)])
        med = sorted(range(n_passes), key=lambda i_: pass_elapsed[i_])[n_passes // 2]      # the MEDIAN pass is the one reported
        elapsed, per_call_ms = pass_elapsed[med], pass_calls[med]
        prof, fw = ctx.profile_read(), ctx.profile_read_forwards()      # (HIP events around the conv launches of ALL passes: means are unaffected)
        ctx.profile_enable(False)
        ctx.set_pruning(True)
        ar_conv_ms = None
        if ar_ctx is not None:
            ar_conv_ms = sum(p_['ms'] for p_ in ar_ctx.profile_read())
            ar_ctx.profile_read_forwards()
            ar_ctx.profile_enable(False)
        for p in final.values():
            assert bool(torch.isfinite(p).all()), 'non-finite pose'
        pool1 = ctx.pool_stats()
        if os.environ.get('DDK_BENCH_TRACE'):
            fr, tot_mem = torch.cuda.mem_get_info(dev)
            print('[bench] bracket done: score pool', pool1, 'conf pool', extra['confidence_model'].ctx.pool_stats() if with_conf else None,
                  'ar pool', extra['ar_model'].score_model.ctx.pool_stats() if disco and hasattr(extra['ar_model'], 'score_model') else None,
                  'device memory in use GB', round((tot_mem - fr) / 2**30, 1), file=sys.stderr, flush=True)
        nres_ = sorted(int(complexes[i]['rec_pos'].shape[0]) for i in mine)
        stream = {'complexes_created': a.steps, 'distinct_complexes': len(mine), 'receptor_residues_min_median_max': [nres_[0], nres_[len(nres_) // 2], nres_[-1]],
                  'ligand_atoms_min_max': [int(min(complexes[i]['lig_pos'].shape[0] for i in mine)), int(max(complexes[i]['lig_pos'].shape[0] for i in mine))],
                  'chunk_pool_hipMalloc_in_timed_region': pool1['hipMalloc_calls'] - pool0['hipMalloc_calls'],
                  'chunk_pool_reuses_in_timed_region': pool1['reuses'] - pool0['reuses'],
                  'chunk_pool_hipFree_in_timed_region': pool1['hipFree_calls'] - pool0['hipFree_calls'],
                  'chunk_pool_bytes_parked': pool1['bytes_parked'], 'complex_bytes_owned_peak': pool1['bytes_owned_peak'],
                  'device_memory_in_use_peak_bytes': int(mem_peak)}
        return dict(elapsed=elapsed, prof=prof, fw=fw, per_call_ms=per_call_ms, final=final, confs=confs, order=order[warmup:warmup + a.steps], stream=stream,
                    passes=n_passes, pass_elapsed_s=[round(v, 5) for v in pass_elapsed], ar_conv_ms=ar_conv_ms)

    layer_flop = [2 * 72 * (72 + W_LAYER[l]) + TP_FLOP[l] for l in range(5)]

    def summary(r, n_units):
        """what a bracket reports beside the headline: rate, conv share, executed-edge fraction"""
        e_x, e_u = sum(p['edges'] for p in r['prof']), sum(p['edges_unpruned'] for p in r['prof'])
        conv_ms = sum(p['ms'] for p in r['prof'])
        fl = sum(p['edges'] * layer_flop[l] for l, p in enumerate(r['prof']))
        fw = r['fw']
        cross = fw[:, 3].reshape(-1, STEPS) / b_local if len(fw) and len(fw) % STEPS == 0 else None
        return {'value': n_units / r['elapsed'], 'unit': 'complexes/s', 'ms_per_step': 1e3 * r['elapsed'] / a.steps,
                'edges_executed_over_unpruned': e_x / max(e_u, 1), 'conv_fp32_equivalent_TFLOPs': fl / max(conv_ms, 1e-9) / 1e9,
                'conv_share_of_wall': conv_ms * 1e-3 / sum(r['pass_elapsed_s']),
                'conv_share_of_wall_incl_ar_model': None if r.get('ar_conv_ms') is None else (conv_ms + r['ar_conv_ms']) * 1e-3 / sum(r['pass_elapsed_s']),
                'min_cross_edges_per_sample_over_steps': None if cross is None else float(cross.min())}

    for _rep in range(int(os.environ.get('DDK_BENCH_REPEAT', '1')) - 1):      # debugging aid: the pocket-bound bracket several times in one process
        print(f'[bench] repeat {_rep}', file=sys.stderr, flush=True)
        bracket({i: pocket_poses(complexes[i], np.random.default_rng(1000 + i), SAMPLES) for i in mine}, a.warmup, noise_scale=0.2)
    if os.environ.get('DDK_BENCH_HEADLINE_POCKET'):      # debugging aid: the pocket-bound workload in the headline's place (with --no-extras: that bracket alone)
        head = bracket({i: pocket_poses(complexes[i], np.random.default_rng(1000 + i), SAMPLES) for i in mine}, a.warmup, noise_scale=0.2)
    else:
        head = bracket(poses_all, a.warmup)
    elapsed, prof, final, confs = head['elapsed'], head['prof'], head['final'], head['confs']

    # ---- the one exchange of the path: final poses (and confidences) of every complex to every rank (RCCL over xGMI) ----------
    if big:
        gathered = {i: gather_samples(p, SAMPLES, rank, world, dev) for i, p in final.items()}
        assert all(g.shape[0] == SAMPLES for g in gathered.values())
        n_done = a.steps                                    # the ranks worked on the SAME K complexes
    else:
        n_total = n_cx if shard_set else n_cx * world
        nl = torch.zeros(n_total, dtype=torch.int64, device=dev)
        for i in mine:
            nl[i] = complexes[i]['lig_pos'].shape[0]
        if use_dist:
            dist.all_reduce(nl)
        pose_digest = None
        if len(final) == len(mine):
            gathered = gather_poses(final, [int(v) for v in nl.tolist()], SAMPLES, dev)
            assert len(gathered) == n_total
            gconf = None
            if with_conf:
                gconf = gather_confidences(confs, n_total, dev)
                assert len(gconf) == n_total
            if shard_set and rank == 0:
                import hashlib
                h = hashlib.sha256()
                for i in sorted(gathered):
                    h.update(gathered[i].cpu().numpy().tobytes())
                pose_digest = {'sha256_of_the_gathered_poses': h.hexdigest(), 'complexes': n_total,
                               'pose_checksum': float(sum(float(gathered[i].double().sum()) for i in gathered)),
                               'confidence_checksum': None if gconf is None else float(sum(float(gconf[i].double().sum()) for i in gconf)),
                               'complexes_per_rank': [len(shard_indices([complex_cost(r_, l_ or 30) for r_, l_ in shapes], r, world)) for r in range(world)],
                               'note': 'DDK_DETERMINISTIC=1: bit-identical for every number of ranks (the confidence model keeps its atomics: its checksum agrees to ~1e-6)'}
        n_done = n_total if shard_set else world * a.steps

    if a.dump_poses and rank == 0:
        np.savez(a.dump_poses, **{f'c{i}': gathered[i].cpu().numpy() for i in sorted(gathered)})

    # ---- what the headline rests on: the same bracket without the receptive-field pruning (the floor) and on a pocket-bound workload ----
    pruning_off = pocket_bound = other_limbs = None
    kern_default = int(getattr(ctx.cfg, 'conv_kernel', 0))
    if not a.no_extras and not a.no_alt and kern_default in (0, 3) and not shard_set:
        # the same sampling() bracket with the OTHER form of the f16-limb conv kernel in every context (score model, AR model, confidence model): conv_kernel = 3
        # (three limbs, six products: the default of rounds 3 - 5) beside the default 0 (two limbs, three products), or the other way round under DDK_CONV_KERNEL=3
        other_k = 3 if kern_default == 0 else 0
        print(f'[bench] extras: conv_kernel = {other_k} bracket', file=sys.stderr, flush=True)
        mdl_o = build_models(conv_kernel=other_k)
        assert int(mdl_o[1].ctx.cfg.conv_kernel) == other_k
        r_o = bracket(poses_all, min(a.warmup, 2), mdl=mdl_o)
        other_limbs = summary(r_o, n_done)
        conv_ms_o, launches_o = sum(p_['ms'] for p_ in r_o['prof']), sum(p_['launches'] for p_ in r_o['prof'])
        nt_o = [len(mdl_o[1].ctx.export(f'conv.{l}.tiles', dtype=np.int32)) // 4 for l in range(5)]
        prod_o = 6 if other_k == 3 else 3
        mfma_o = sum(p_['edges'] * MFMA_FLOP_PER_EDGE_TILE[prod_o] * (nt_o[l] + 1) for l, p_ in enumerate(r_o['prof']))
        other_limbs.update(conv_kernel=other_k, limb_products=prod_o, avg_launch_ms=conv_ms_o / max(launches_o, 1),
                           mfma_TFLOPs_executed=mfma_o / max(conv_ms_o, 1e-9) / 1e9, frac_of_f16_matrix_peak=mfma_o / max(conv_ms_o, 1e-9) / 1e9 / PEAK_F16_MFMA_TFLOPS,
                           note=('the same bracket, workload, seeds and pruning with ddk_config.conv_kernel = %d in every context: ' % other_k) +
                                ('three fp16 limbs per operand, six limb products exact to 2^-33 (k_conv_x.hip: the default of ddk 0.4 - 0.7)' if other_k == 3 else
                                 'two fp16 limbs per operand, three limb products (k_conv_x2.hip: the default)'))
        del mdl_o, r_o
    if not a.no_extras:
        w2 = min(a.warmup, 2)
        print('[bench] extras: pruning_off bracket', file=sys.stderr, flush=True)
        pruning_off = summary(bracket(poses_all, w2, prune=False), n_done)
        pruning_off['note'] = ('the same sampling() bracket, workload and noise with ddk_set_receptive_field_pruning(ctx, 0): every layer evaluates all '
                               'rec-rec messages (layer-0 de-duplication and the last layer\'s ligand-only evaluation stay) - the guaranteed floor of value')
        pk_poses = {i: pocket_poses(complexes[i], np.random.default_rng(1000 + i), SAMPLES) for i in mine}
        print('[bench] extras: pocket_bound bracket', file=sys.stderr, flush=True)
        pocket_bound = summary(bracket(pk_poses, w2, noise_scale=0.2), n_done)
        print('[bench] extras done', file=sys.stderr, flush=True)
        pocket_bound['note'] = ('the same bracket with the pruning ON, start poses inside the pocket (rotation about the centroid + N(0, 1 A)) and the N(0,1) '
                                'draws scaled by 0.2 (pre-drawn per call): every sample keeps cross edges for all 20 steps, as the trajectories of a '
                                'trained model do; the default workload starts from randomize_position (N(0, 19 A)) and random-init weights let the '
                                'ligand wander')

    # ---- host cost of ddk_complex_create (topology + staged uploads; the ESM projection and the receptor-edge terms run on the upload stream) ------
    create_ms = None
    if rank == 0 and not a.no_extras:
        create_ms = {}
        for nr in sorted({n_res, 300, 2000}):
            cc = complexes[mine[0]] if nr == n_res else synthetic.make_complex(0, n_res=nr)
            ts = []
            for rep in range(6):
                torch.cuda.synchronize()
                t_c = time.perf_counter()
                cxc = Complex(ctx, cc, b_local)
                ts.append(time.perf_counter() - t_c)
                torch.cuda.synchronize()
                cxc.close()
            create_ms[f'{nr}_residues'] = round(1e3 * float(np.median(ts[1:])), 3)

    # ---- comparison figure: the loop alone on resident complexes with pre-drawn noise (round 1's bracket) ----------------------
    device_loop = None
    if not a.no_device_loop and not disco:
        cxs = {i: Complex(ctx, complexes[i], b_local) for i in mine}
        p0 = {i: torch.from_numpy(poses_all[i][lo:hi]).to(dev) for i in mine}
        torch.cuda.manual_seed(4321 + rank)      # the same draws, in the same order, as the timed sampling() calls above: identical work
        nz = {k: draw_noise(STEPS, b_local, cxs[order[k]].R, cxs[order[k]].R, nc, dev) for k in range(a.warmup, a.warmup + a.steps)}
        for k in range(a.warmup):
            cxs[order[k]].sample(p0[order[k]].clone(), t_arr, sc, nc, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(a.warmup, a.warmup + a.steps):
            cxs[order[k]].sample(p0[order[k]].clone(), t_arr, sc, nc, nz[k])
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        device_loop = {'value': (1 if big else world) * a.steps / e1, 'unit': 'complexes/s', 'ms_per_step': 1e3 * e1 / a.steps,
                       'note': 'complexes, start poses and noise resident in HBM before the clock starts (this rank; round-1 bracket)'}
        del cxs

    if rank == 0:
        conv_ms = sum(p['ms'] for p in prof)
        fl = lambda key: sum(p[key] * layer_flop[l] for l, p in enumerate(prof))
        flops_exec, flops_unpruned, flops_full = fl('edges'), fl('edges_unpruned'), fl('edges_reference')
        n_tiles = [len(ctx.export(f'conv.{l}.tiles', dtype=np.int32)) // 4 for l in range(5)]        # W2 tiles of 32 rows per layer (59 for W = 1872)
        products = 6 if int(getattr(ctx.cfg, 'conv_kernel', 0)) == 3 else 3
        mfma_tile = MFMA_FLOP_PER_EDGE_TILE[products]
        mfma_exec = sum(p['edges'] * mfma_tile * (n_tiles[l] + 1) for l, p in enumerate(prof))
        byts = sum(p['edges'] * FUSED_BYTES[l] for l, p in enumerate(prof))
        launches = sum(p['launches'] for p in prof)
        tf = lambda f: f / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        traffic, traffic_file = pmc_traffic()
        if not (a.config == 2 and a.steps == 20 and a.warmup == 5 and not a.complexes and world == 1):
            # the PMC passes are taken over the driver's command: their per-launch mean is that workload's, not another config's
            traffic, traffic_file = None, 'the PMC profile is of the driver command (config 2, --steps 20 --warmup 5): not quoted for another workload'
        fw = head['fw']
        per_step = None
        if len(fw) == head['passes'] * a.steps * STEPS:
            f3 = fw.reshape(head['passes'] * a.steps, STEPS, 4)
            per_step = [{'step': s_, 't': round(float(t_arr[s_, 0]), 3), 'edges_executed_over_unpruned': float(f3[:, s_, 1].sum() / max(f3[:, s_, 2].sum(), 1)),
                         'conv_ms': float(f3[:, s_, 0].mean()), 'cross_edges_per_sample': float(f3[:, s_, 3].mean() / b_local)} for s_ in range(STEPS)]
        # device time of the timed calls that ran the SAME complex (same ligand size, own noise): max / min - 1 over each complex' calls
        by_cx = {}
        for i, ms in zip(head['order'], head['per_call_ms']):
            by_cx.setdefault(i, []).append(ms)
        rep = [max(v) / min(v) - 1.0 for v in by_cx.values() if len(v) > 1]
        per_call_spread = round(max(rep), 4) if rep else None
        # device ms per complex by receptor-size decile (the timesplit-shaped set: cost grows with the receptor, distributed.complex_cost is fitted to this)
        per_decile = None
        if len(set(int(complexes[i]['rec_pos'].shape[0]) for i in mine)) > 1:
            rows_ = sorted((int(complexes[i]['rec_pos'].shape[0]), int(complexes[i]['lig_pos'].shape[0]), ms) for i, ms in zip(head['order'], head['per_call_ms']))
            per_decile = []
            for d_ in range(10):
                part = rows_[len(rows_) * d_ // 10:len(rows_) * (d_ + 1) // 10]
                if part:
                    per_decile.append({'decile': d_, 'residues': [part[0][0], part[-1][0]], 'complexes': len(part), 'mean_ligand_atoms': round(float(np.mean([r_[1] for r_ in part])), 1),
                                       'ms_per_complex': round(float(np.mean([r_[2] for r_ in part])), 2)})
        tp_boundary = None
        if world == 1 and cfg_id == 2 and not a.no_tp_boundary and not a.no_extras:
            tp_boundary = tp_boundary_a(dev)
        out = {
            'metric': 'complexes/sec, 20-step 40-sample inference',
            'value': n_done / elapsed, 'unit': 'complexes/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            # the three figures belong together: value rests on where random-init weights push the ligand; value_pocket_bound is what a trained
            # checkpoint's trajectories look like (every sample keeps >= 2 500 cross edges for all 20 steps), value_pruning_off the guaranteed floor of value
            'value_pocket_bound': pocket_bound['value'] if pocket_bound else None, 'value_pruning_off': pruning_off['value'] if pruning_off else None,
            # the same bracket with the three-limb / six-product form of the conv kernel (ddk_config.conv_kernel = 3, the default of rounds 3 - 5) in every context
            'value_six_limb_products': (other_limbs['value'] if (other_limbs and other_limbs['conv_kernel'] == 3) else (n_done / elapsed if kern_default == 3 else None)),
            'ms_per_step': 1e3 * elapsed / a.steps, 'higher_is_better': True, 'scaling': 'strong' if (big or shard_set) else 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'dtype_note': ('every input, weight, accumulator and output of the path is fp32; the radial-MLP GEMMs multiply the fp32 operands on the f16 matrix pipe as ' +
                           ('three f16 limbs each (six of nine limb products, dropped terms <= 3 * 2^-33 relative: ddk_config.conv_kernel = 3)' if kern_default == 3 else
                            'two f16 limbs each, hi + mid rounded to nearest (|x - hi - mid| <= 2^-22 |x|), the three limb products hi.hi + hi.mid + mid.hi (mid.mid <= 2^-22 dropped): '
                            '<= 3 * 2^-22 relative per product, below the classical 72 * 2^-24 of an fp32 dot product of K = 72; measured level with fp32 FMA chains and with the six-product '
                            'form against the fp64 oracle (6 - 10e-8 relative per conv layer, tests/test_gpu_round6.py; value_six_limb_products is the same bracket with conv_kernel = 3)') +
                           ', fp32 accumulation (DESIGN.md 3.3)'),
            'config': {'workload': f'BASELINE config {cfg_id}: ' + CONFIG_TEXT[cfg_id] + ('; receptor sizes of the set drawn timesplit-shaped (synthetic.timesplit_shape: log-normal, '
                                   'median 350, clipped to [60, 3000] residues), ligands 10-80 atoms' if (spread_ligands and not a.fixed_receptor) else '') + '; 1 step = 1 complex',
                       'bracket': 'wall time around sampling(data_list, model, ...) on host data_lists, a new complex every call (evaluate.py:259,293): '
                                  'collation, ddk_complex_create, H2D, noise draws, the 20-step loop, pose write-back; K calls + one final synchronisation; '
                                  f'{head["passes"]} such passes over the same K calls, the MEDIAN pass is reported (value, ms_per_step; pass times in extra.headline.pass_elapsed_s)',
                       # the three figures belong together, here too because the driver's record keeps `config`
                       'value_headline_pruning_off_pocket_bound': [round(n_done / elapsed, 3), None if not pruning_off else round(pruning_off['value'], 3),
                                                                   None if not pocket_bound else round(pocket_bound['value'], 3)],
                       'conv_kernel': kern_default,
                       'value_six_limb_products_conv_kernel_3': None if not (other_limbs and other_limbs['conv_kernel'] == 3) else round(other_limbs['value'], 3),
                       'samples_per_complex': SAMPLES, 'inference_steps': STEPS, 'complexes_per_gpu': n_cx,
                       'parallelism': (f'the {SAMPLES} samples of every complex sharded over {world} process(es) ({b_local} per GPU), final all_gather'
                                       if big else f'{world} process(es), one per GPU, each with the same {n_cx} complexes (own start poses and noise: per-GPU work fixed), final RCCL all_gather of the poses')},
            'roofline': {'bound': 'mfma', 'kernel': ('ddk::conv_x3_kernel<true, true, false> (k_conv_x.hip: fp32 operands as three exact f16 limbs, six limb '
                                                     'products on v_mfma_f32_32x32x16_f16, fp32 accumulators)' if kern_default == 3 else
                                                     'ddk::conv_x2_kernel<true, true, false> (k_conv_x2.hip = k_conv_x.hip with two f16 limbs per fp32 operand, three limb '
                                                     'products on v_mfma_f32_32x32x16_f16, one fp32 accumulator)'),
                         'achieved': tf(mfma_exec), 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf(mfma_exec) / PEAK_F16_MFMA_TFLOPS,
                         'accounting': 'achieved / frac: MFMA FLOPs the launches EXECUTED (per evaluated edge and layer: (W2 tiles + 1 GEMM1) x (' + ('432 K-columns of 32x32 f16 MFMA = 24 x 32x32x16 + 6 x '
                                       '32x32x8 unpacked, per 32 edges) = six' if products == 6 else '224 K-columns of 32x32 f16 MFMA = 14 x 32x32x16 per 32 edges) = three') +
                                       ' limb products of K = 72, rows padded to 32-row tiles) / HIP-event time of the '
                                       'launches, against the dense f16 matrix peak.  fp32_equivalent_TFLOPs: the ALGORITHMIC fp32 FLOPs of the same edges '
                                       '(2*72*(72+W) + TP per edge and layer, BASELINE.md section 3) / the same time - what an fp32 kernel would have to sustain; '
                                       'the fp32 MFMA peak is 157.3.  reference_equivalent_TFLOPs additionally counts the receptor-receptor messages the backward '
                                       'receptive-field pruning proved dead (not a roofline figure: no frac); full_reference_TFLOPs every edge of the reference graph',
                         'fp32_equivalent_TFLOPs': tf(flops_exec), 'fp32_mfma_peak_TFLOPs': PEAK_F32_MFMA_TFLOPS,
                         'fp32_equivalent_over_fp32_mfma_peak': tf(flops_exec) / PEAK_F32_MFMA_TFLOPS,
                         'reference_equivalent_TFLOPs': tf(flops_unpruned), 'full_reference_TFLOPs': tf(flops_full),
                         'edges_executed_over_unpruned': sum(p['edges'] for p in prof) / max(sum(p['edges_unpruned'] for p in prof), 1),
                         'traffic': traffic,
                         'traffic_source': (f'profiles/{traffic_file} (taken on this kernel: source sha256 {conv_kernel_source_sha()[:16]}): rocprofv3 --pmc '
                                            'FETCH_SIZE / WRITE_SIZE (separate passes) over this command, bytes per conv launch = 2*FETCH_SIZE + WRITE_SIZE '
                                            '(gfx950 FETCH correction)') if traffic is not None else f'null: {traffic_file}',
                         'algorithmic_bytes_per_launch': byts / max(launches, 1),
                         'launches': launches, 'avg_launch_ms': conv_ms / max(launches, 1),
                         'flop_per_launch': mfma_exec / max(launches, 1), 'fp32_equivalent_flop_per_launch': flops_exec / max(launches, 1),
                         'algorithmic_hbm_GBps': byts / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0,
                         'algorithmic_hbm_frac_of_peak': (byts / (conv_ms * 1e-3) / 1e9) / PEAK_HBM_GBS if conv_ms > 0 else 0.0,
                         'conv_share_of_wall': conv_ms * 1e-3 / sum(head['pass_elapsed_s']),
                         # configs 3 / 4: + the conv launches of the AR latent model's two encoder passes per complex (its own context; the all-atom confidence
                         # model's nine-group launches are not event-timed)
                         'conv_share_of_wall_incl_ar_model': None if head.get('ar_conv_ms') is None else (conv_ms + head['ar_conv_ms']) * 1e-3 / sum(head['pass_elapsed_s']),
                         # the BASELINE metric's second clause at the REFERENCE's op boundary (tensor_layers.py:65-116, weights [E, W] in HBM): the HBM-bound kernel
                         'tp_boundary_A': tp_boundary,
                         'limb_products': products,
                         # the other form of the f16-limb kernel through the same bracket (None: --no-alt / --no-extras): what the executed-MFMA fraction is when six products are multiplied
                         'other_limb_form': None if not other_limbs else {k_: other_limbs[k_] for k_ in ('conv_kernel', 'limb_products', 'value', 'avg_launch_ms', 'mfma_TFLOPs_executed',
                                                                                                          'frac_of_f16_matrix_peak', 'conv_fp32_equivalent_TFLOPs')},
                         'per_layer': [{'layer': l, 'ms_per_launch': p['ms'] / max(p['launches'], 1), 'w2_tiles': n_tiles[l],
                                        'mfma_TFLOPs': p['edges'] * mfma_tile * (n_tiles[l] + 1) / max(p['ms'], 1e-9) / 1e9,
                                        'fp32_equivalent_TFLOPs': p['edges'] * layer_flop[l] / max(p['ms'], 1e-9) / 1e9,
                                        'edges_executed_frac': p['edges'] / max(p['edges_unpruned'], 1)}
                                       for l, p in enumerate(prof)]},
            'extra': {'pruning_off': pruning_off, 'pocket_bound': pocket_bound, 'other_limb_form': other_limbs, 'per_step': per_step, 'create_ms': create_ms, 'pose_digest': None if big else pose_digest,
                      'per_call_spread_same_complex': per_call_spread,
                      'headline': dict({k: v for k, v in summary(head, n_done).items() if k != 'value'}, passes=head['passes'], pass_elapsed_s=head['pass_elapsed_s']),
                      'per_receptor_size_decile': per_decile,
                      'device_loop': device_loop, 'per_call_ms': head['per_call_ms'] if a.steps <= 64 else head['per_call_ms'][:64] + ['...'], 'stream': head['stream']},
        }
        if int(getattr(ctx.cfg, 'conv_kernel', 0)) == 1:
            # the whole run was switched to the fallback kernel (DDK_CONV_KERNEL=1): its work is fp32 MFMA chains, priced against the fp32 MFMA peak
            r_ = out['roofline']
            r_['kernel'] = 'ddk::conv_fused_kernel<true, 0> (k_conv.hip: radial-MLP GEMMs as v_mfma_f32_32x32x2_f32 chains; the fallback, ddk_config.conv_kernel = 1)'
            r_['achieved'] = r_['fp32_equivalent_TFLOPs']
            r_['peak'] = PEAK_F32_MFMA_TFLOPS
            r_['frac'] = r_['fp32_equivalent_TFLOPs'] / PEAK_F32_MFMA_TFLOPS
            r_['accounting'] = ('achieved / frac: the ALGORITHMIC fp32 FLOPs of the evaluated edges (2*72*(72+W) + TP per edge and layer) / HIP-event time of the launches, '
                                'against the fp32 MFMA peak (the kernel executes them as fp32 MFMA chains, K padded 72 -> 80)')
            for k_ in ('flop_per_launch',):
                r_[k_] = r_['fp32_equivalent_flop_per_launch']
            for pl in r_['per_layer']:
                pl.pop('mfma_TFLOPs', None)
            r_['traffic'] = None
            r_['traffic_source'] = None
            out['dtype_note'] = 'every operand and accumulator of the path is fp32 (fallback kernel: fp32 MFMA chains)'
        if world == 1 and not a.no_cpu_baseline and not disco:
            c0 = complexes[mine[0]]
            cx0 = Complex(ctx, c0, SAMPLES)

            def gpu_scores(pos, t):
                return cx0.score_forward(torch.from_numpy(pos).to(dev), t, t, t)
            out['cpu_baseline'] = cpu_baseline(c0, P, coeffs, gpu_scores, n_res, poses_all[mine[0]], chunk=8 if n_res <= 300 else 2)
        else:
            out['cpu_baseline'] = None
        if world == 1 and cfg_id == 2 and not a.no_alt:
            # the stated fallback ddk_config.conv_kernel = 1 (radial-MLP GEMMs as fp32 MFMA chains, k_conv.hip): the same resident loop
            from disco_diffdock_amd.runtime import Context
            ctx2 = Context(device=local, conv_kernel=1)
            ctx2.load_state_dict(P)
            cxs = {i: Complex(ctx2, complexes[i], SAMPLES) for i in mine}
            gen = torch.Generator(device=dev).manual_seed(1234)
            p0 = {i: torch.from_numpy(poses_all[i]).to(dev) for i in mine}
            nz = {i: torch.randn((STEPS, SAMPLES, 6 + cxs[i].R), device=dev, generator=gen) for i in mine}
            for k in range(a.warmup):
                cxs[order[k]].sample(p0[order[k]].clone(), t_arr, sc, nc, nz[order[k]])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for k in range(a.warmup, a.warmup + a.steps):
                cxs[order[k]].sample(p0[order[k]].clone(), t_arr, sc, nc, nz[order[k]])
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t2
            out['fallback_fp32_kernel'] = {'mode': 'ddk_config.conv_kernel = 1: v_mfma_f32_32x32x2_f32 chains (k_conv.hip), resident-loop bracket (compare with '
                                                   'extra.device_loop)',
                                           'value': a.steps / e2, 'unit': 'complexes/s', 'ms_per_step': 1e3 * e2 / a.steps}
        if world == 1 and cfg_id == 2 and not a.no_extras and not a.no_timesplit and a.complexes == 0:
            out['extra']['timesplit_stream'] = timesplit_stream()
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
