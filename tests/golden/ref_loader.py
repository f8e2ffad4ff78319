"""Import the reference's model files from /root/reference WITHOUT importing the `live2diff`
package (its __init__ pulls in the real diffusers / omegaconf / MiDaS, absent here).

TEST INFRASTRUCTURE ONLY -- used by gen_golden.py in the build container.  /root/reference does
not exist on the GPU box; nothing under -m gpu, smoke() or bench.py may import this module.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("L2D_REFERENCE_ROOT", "/root/reference")
_MODELS = os.path.join(REF_ROOT, "live2diff", "animatediff", "models")
_PKG = "l2dref_models"


def available():
    return os.path.isdir(_MODELS)


def load():
    """Returns a namespace with the reference modules (resnet, attention, motion_module, ...)."""
    if _PKG in sys.modules and hasattr(sys.modules[_PKG], "_loaded"):
        return sys.modules[_PKG]
    sys.path.insert(0, os.path.dirname(__file__))
    import diffusers_stub

    diffusers_stub.install()
    pkg = types.ModuleType(_PKG)
    pkg.__path__ = [_MODELS]
    sys.modules[_PKG] = pkg
    order = [
        "resnet",
        "positional_encoding",
        "attention",
        "stream_motion_module",
        "motion_module",
        "unet_blocks_streaming",
        "unet_blocks_warmup",
        "unet_depth_streaming",
        "unet_depth_warmup",
    ]
    for name in order:
        spec = importlib.util.spec_from_file_location(f"{_PKG}.{name}", os.path.join(_MODELS, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"{_PKG}.{name}"] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
    pkg._loaded = True
    return pkg


def load_pipeline_class():
    """The reference's StreamAnimateDiffusionDepth class object (methods used unbound on a fake self
    to capture the ring-buffer state machine and the LCM step; nothing heavy is instantiated)."""
    load()
    import torch

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    d = sys.modules["diffusers"]
    d.LCMScheduler = _Dummy
    mod("diffusers.image_processor", VaeImageProcessor=_Dummy)
    mod("diffusers.pipelines")
    mod("diffusers.pipelines.stable_diffusion")
    mod("diffusers.pipelines.stable_diffusion.pipeline_stable_diffusion_img2img", retrieve_latents=lambda *a, **k: None)
    lp = types.ModuleType("l2dref_pkg")
    lp.__path__ = []
    sys.modules["l2dref_pkg"] = lp
    mod("live2diff")
    mod("live2diff.image_filter", SimilarImageFilter=_Dummy)
    ad = mod("l2dref_pkg.animatediff")
    ad.__path__ = []
    mod("l2dref_pkg.animatediff.pipeline", AnimationDepthPipeline=_Dummy)
    path = os.path.join(REF_ROOT, "live2diff", "pipeline_stream_animation_depth.py")
    spec = importlib.util.spec_from_file_location("l2dref_pkg.pipeline_stream_animation_depth", path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    return m
