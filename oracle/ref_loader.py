"""Imports the reference's hot-path modules from /root/reference WITHOUT running kandinsky2/__init__.py
(which needs a removed huggingface_hub API and omegaconf) and without modifying the reference tree.
Only usable in the build container; the GPU box has no /root/reference (SURVEY.md Appendix B recipe).
"""
import importlib
import os
import sys
import types

import torch.nn as nn

REF_ROOT = os.environ.get("K22_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "kandinsky2", "model"))


def ref(leaf: str):
    """ref('model.unet'), ref('model.model_creation'), ref('vqgan.autoencoder'), ref('configs') ..."""
    if not available():
        raise RuntimeError("reference tree not present")
    if "kandinsky2" not in sys.modules:
        for name, path in [("kandinsky2", ""), ("kandinsky2.model", "/model"), ("kandinsky2.vqgan", "/vqgan")]:
            m = types.ModuleType(name)
            m.__path__ = [REF_ROOT + "/kandinsky2" + path]
            sys.modules[name] = m
        pl = types.ModuleType("pytorch_lightning")  # autoencoder.py:3 (import-time only)
        pl.LightningModule = nn.Module
        sys.modules.setdefault("pytorch_lightning", pl)
        clip = types.ModuleType("clip")  # prior.py:10-12 (tokenizer base class, import-time only)
        st = types.ModuleType("clip.simple_tokenizer")
        st.SimpleTokenizer = type("SimpleTokenizer", (), {})
        st.default_bpe = lambda: None
        clip.simple_tokenizer = st
        sys.modules.setdefault("clip", clip)
        sys.modules.setdefault("clip.simple_tokenizer", st)
    return importlib.import_module("kandinsky2." + leaf)
