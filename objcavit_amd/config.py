"""Configuration objects for the drop-in modules.

The reference passes an OmegaConf ``DictConfig`` called ``args`` into every
module (main.py:161-163).  The hot path only ever does ``args.a.b``,
``args["a"]``, ``args[args.model.name]`` and ``.get("key")`` on it
(modules/ObjCAViT.py:35-36,159,244,293; modules/GraphBins.py:45-52,111;
modules/AdaBins.py:33,43,79), so any mapping with attribute access works --
an OmegaConf object from the reference's own main.py can be passed straight in.
``AttrDict`` is that mapping for users without OmegaConf, ``load_yaml`` reads
the reference's params/*.yaml files into it, and ``make_args`` builds the
handful of knobs the path reads for the synthetic benchmark configurations.
"""
from __future__ import annotations

from typing import Any, Mapping


class AttrDict(dict):
    """dict with attribute access, recursive over nested mappings."""

    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, Mapping) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, list):
            return [AttrDict._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def load_yaml(path: str) -> AttrDict:
    """Read one of the reference's params/*.yaml files (main.py:161)."""
    import yaml
    with open(path) as f:
        return AttrDict(yaml.safe_load(f))


# dataset constants of params/basicParams.yaml:109-160
_DATASETS = {
    "nyu": dict(min_depth=0.001, max_depth=10, dimensions_train=[416, 544], dimensions_test=[480, 640],
                do_kb_crop=False, eigen_crop=True, garg_crop=False),
    "kitti": dict(min_depth=0.001, max_depth=80, dimensions_train=[352, 704], dimensions_test=[376, 1241],
                  do_kb_crop=True, eigen_crop=False, garg_crop=True),
}


def load_reference_config(path: str, basic_params: str = None) -> AttrDict:
    """A run's ``params/*.yaml`` as the reference sees it in validate / inference mode (main.py:161-179 ->
    misc_utils.check_and_validate_args, misc_utils.py:40-48): the file itself, with its ``nyu`` and ``kitti`` blocks
    REPLACED by those of ``params/basicParams.yaml`` ("those params never change").  ``basic_params`` is that file's
    path; without it the same constants come from this module's table (basicParams.yaml:109-160, the keys the hot
    path reads)."""
    args = load_yaml(path)
    if basic_params is not None:
        base = load_yaml(basic_params)
        args["nyu"], args["kitti"] = base["nyu"], base["kitti"]
    else:
        for name, consts in _DATASETS.items():
            blk = AttrDict(args.get(name) or {})
            blk.update(consts)
            args[name] = blk
    return args


def make_args(model: str = "graphbins", dataset: str = "nyu", *, strategy: str = "learned",
              embedding_dim: int = 128, language: str = "control_obj_zeros_512", no_obj_sa: bool = False,
              use_2_saca: bool = False, n_bins: int = 256, encoder_name: str = "efficientnet-b5",
              do_final_upscale: bool = False, **dataset_overrides: Any) -> AttrDict:
    """The subset of the reference's YAML tree that the hot path reads."""
    ds = dict(_DATASETS[dataset])
    ds.update(dataset_overrides)
    objcavit = dict(positional_embedding_strategy=strategy, embedding_dim=embedding_dim,
                    language_embedding_strategy=language, obj_language_strategy="none")
    if no_obj_sa:
        objcavit["no_obj_sa"] = True
    if use_2_saca:
        objcavit["use_2_saca"] = True
    block = dict(n_bins=n_bins, encoder_name=encoder_name)
    if do_final_upscale:
        block["do_final_upscale"] = True
    args = AttrDict(basic=dict(dataset=dataset), model=dict(name=model))
    args["graphbins"] = dict(block, objcavit=objcavit, yolov7_chkpt="./yolov7_chkpts/yolov7-seg-lvis-e234.pt")
    args["adabins"] = dict(block)
    args[dataset] = ds
    return args
