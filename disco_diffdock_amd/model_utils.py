"""Host mirror of utils/model_utils.py:25-101 (get_model): model_parameters.yml Namespace -> score model."""
from functools import partial

import torch

from .diffusion_utils import get_timestep_embedding, t_to_sigma as t_to_sigma_compl
from .score_model import TensorProductScoreModel, ModelWrapper


def get_model(args, device, t_to_sigma, no_parallel=False, confidence_mode=False):
    if getattr(args, 'all_atoms', False):
        if not confidence_mode:
            raise RuntimeError('ddk: the all-atom model is implemented in confidence_mode only (the all-atom SCORE model is outside the hot path)')
        from .confidence import ConfidenceModel       # utils/model_utils.py:26-27 -> AAScoreModel
        return ConfidenceModel(args, device)
    g = lambda k, d: getattr(args, k, d)
    if g('latent_dim', 0) > 0 and g('latent_vocab', 0) != 1:
        raise RuntimeError('ddk: latent conditioning is implemented for the equivariant-latent models (latent_vocab == 1) only')
    lm = 'esm' if g('esm_embeddings_path', None) is not None else None
    score_model = TensorProductScoreModel(
        t_to_sigma=t_to_sigma, device=device, no_torsion=args.no_torsion,
        timestep_emb_func=get_timestep_embedding(args.embedding_type, args.sigma_embed_dim, args.embedding_scale),
        num_conv_layers=args.num_conv_layers, lig_max_radius=args.max_radius, scale_by_sigma=args.scale_by_sigma,
        sigma_embed_dim=args.sigma_embed_dim, ns=args.ns, nv=args.nv, distance_embed_dim=args.distance_embed_dim,
        cross_distance_embed_dim=args.cross_distance_embed_dim, batch_norm=not args.no_batch_norm, dropout=args.dropout,
        sh_lmax=g('sh_lmax', 2), use_second_order_repr=args.use_second_order_repr, cross_max_distance=args.cross_max_distance,
        dynamic_max_cross=args.dynamic_max_cross, lm_embedding_type=lm, confidence_mode=confidence_mode,
        use_old_atom_encoder=g('use_old_atom_encoder', True), latent_dim=g('latent_dim', 0), latent_vocab=g('latent_vocab', 0),
        latent_cross_attention=g('latent_cross_attention', False), latent_droprate=g('latent_droprate', 0),
        embedding_scale=args.embedding_scale,
        num_confidence_outputs=len(g('rmsd_classification_cutoff', None)) + 1 if isinstance(g('rmsd_classification_cutoff', None), list) else 1,
        confidence_no_batchnorm=g('confidence_no_batchnorm', False), confidence_dropout=g('confidence_dropout', 0),
        sigma_limits=dict(tr_sigma_min=args.tr_sigma_min, tr_sigma_max=args.tr_sigma_max, rot_sigma_min=args.rot_sigma_min,
                          rot_sigma_max=args.rot_sigma_max, tor_sigma_min=args.tor_sigma_min, tor_sigma_max=args.tor_sigma_max))
    if hasattr(args, 'latent_vocab'):
        return ModelWrapper(encoder=None, score_model=score_model)   # model_utils.py:93-94
    return score_model


def get_ar_model(args, score_model_args, device, training=True):
    """utils/model_utils.py:104-152 for `use_pretrained_score: true` AR models (the shipped disco_diffdockS_ar_model).
    The checkpoint of the ORIGINAL score model (args.original_model_dir/args.ckpt) is not needed at inference: the AR
    checkpoint overwrites it (evaluate.py:179-180), so it is only loaded when present."""
    if training:
        raise RuntimeError('ddk: AR model training is outside the accelerated hot path (training=False only)')
    if not ('use_pretrained_score' in args and args.use_pretrained_score):
        raise RuntimeError('ddk: only AR models built on the pretrained score model (use_pretrained_score) are implemented')
    from .pretrained_score_encoder import PretrainedScoreEncoder
    t_to_sigma = partial(t_to_sigma_compl, args=score_model_args)
    model = get_model(score_model_args, device, t_to_sigma)
    score_model = getattr(model, 'score_model', model)
    ar = PretrainedScoreEncoder(pretrained_score_model=score_model, ns=args.ns, latent_dim=1,
                                latent_vocab=score_model_args.latent_vocab, latent_no_batchnorm=args.latent_no_batchnorm,
                                latent_dropout=args.latent_dropout, latent_hidden_dim=args.latent_hidden_dim,
                                input_latent_dim=score_model_args.latent_dim, apply_gumbel_softmax=True)
    return ar.to(device)
