"""yaml `target:` / `params:` config surface (reference diffmodels/base_diffusion_makeup.yaml, resolved upstream by
cldm.model.create_model / ldm.util.instantiate_from_config at runs/test.py:27).  PyYAML SafeLoader only."""
from __future__ import annotations

import importlib
from typing import Any, Dict

import torch
import yaml

# reference / upstream dotted names -> where this package implements them
TARGET_ALIASES = {
    'diffmk.diffusion_makeup.BaseDoubleControlModel': 'makeupdiffuse_amd.diffmk.diffusion_makeup.BaseDoubleControlModel',
    'diffmk.diffusion_makeup.TestDoubleControlModel': 'makeupdiffuse_amd.diffmk.diffusion_makeup.TestDoubleControlModel',
    'diffmk.makeup_diffuse.BaseMakeUpDiffuse': 'makeupdiffuse_amd.diffmk.makeup_diffuse.BaseMakeUpDiffuse',
    'diffmk.makeup_diffuse.TestDiffuseModel': 'makeupdiffuse_amd.diffmk.makeup_diffuse.TestDiffuseModel',
    'diffmk.makeups.BaseModel': 'makeupdiffuse_amd.diffmk.makeups.BaseModel',
    'diffmk.makeup_controlnet.MakeupDoubleControlModel': 'makeupdiffuse_amd.diffmk.makeup_controlnet.MakeupDoubleControlModel',
}


def load_yaml(path: str) -> Dict[str, Any]:
    with open(path, 'r') as f:
        return yaml.load(f, Loader=yaml.SafeLoader)


def get_obj_from_str(name: str):
    name = TARGET_ALIASES.get(name, name)
    module, cls = name.rsplit('.', 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config: Dict[str, Any]):
    if 'target' not in config:
        raise KeyError('Expected key `target` to instantiate.')
    return get_obj_from_str(config['target'])(**dict(config.get('params', {}) or {}))


def create_model(config_path: str):
    """cldm.model.create_model: yaml -> model object (on the host; call .cuda() to build the engine)."""
    cfg = load_yaml(config_path)
    model = instantiate_from_config(cfg['model'])
    return model


def load_state_dict(ckpt_path: str, location: str = 'cpu') -> Dict[str, torch.Tensor]:
    """cldm.model.load_state_dict for .safetensors and tensor-only .ckpt/.pth files.  Pickled checkpoints are
    opened with weights_only=True (nothing from the file is executed)."""
    if ckpt_path.endswith('.safetensors'):
        from safetensors.torch import load_file
        sd = load_file(ckpt_path, device=location)
    else:
        sd = torch.load(ckpt_path, map_location=location, weights_only=True)
    return sd.get('state_dict', sd)
