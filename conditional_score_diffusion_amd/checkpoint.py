"""Reader for the reference's Lightning checkpoints.

A ``.ckpt`` written by the reference's trainer is a torch pickle ``{'state_dict': {...}, 'hyper_parameters': ...}``
whose network weights live under ``score_model.`` with the ``all_modules.{i}.…`` keys this package's adapters use
(lightning_modules/BaseSdeGenerativeModel.py:21 builds ``self.score_model = mutils.create_model(config)``); the
VS-CMDE module also registers the buffers ``sigma_max_y`` / ``sigma_min_y`` that its ``configure_sde`` reads
(lightning_modules/ConditionalSdeGenerativeModel.py:25-40,140-141).  Only tensors are read; nothing is executed.
"""
import torch


def split_lightning_state_dict(ckpt, prefix='score_model.'):
    """-> (network state_dict with the prefix stripped, {other tensor entries})."""
    sd = ckpt.get('state_dict', ckpt)
    net, rest = {}, {}
    for k, v in sd.items():
        if k.startswith(prefix):
            net[k[len(prefix):]] = v
        else:
            rest[k] = v
    if not net:
        raise KeyError('no %r entries in the checkpoint (keys start with: %s)' % (prefix, sorted({k.split(".")[0] for k in sd})))
    return net, rest


def load_lightning_checkpoint(model, path_or_ckpt, prefix='score_model.', strict=True):
    """Load the network weights of a reference Lightning checkpoint into ``model`` (an adapter of this package).
    Returns the remaining entries (e.g. ``sigma_max_y``), which the caller feeds to its SDE objects."""
    ckpt = path_or_ckpt
    if not isinstance(ckpt, dict):
        ckpt = torch.load(path_or_ckpt, map_location='cpu', weights_only=False)
    net, rest = split_lightning_state_dict(ckpt, prefix)
    model.load_state_dict(net, strict=strict)
    return rest
