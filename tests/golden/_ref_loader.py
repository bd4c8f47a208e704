"""Loader for the read-only reference (used ONLY by gen_golden.py in the build container).

Never imported by tests, bench or the product: /root/reference does not exist on the GPU box.
The four shims are the ordinary import/syntax fixes listed in SURVEY.md §8(c).
"""
import sys, types, re, typing, importlib

REF = "/root/reference"


def load_reference():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import torch
    # (2) torchaudio absent: empty stub (only rnnt_loss in training uses it)
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        ta.functional = types.ModuleType("torchaudio.functional")
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.functional"] = ta.functional
    # (3) whisper absent: wenet/utils/common.py:24 imports LANGUAGES
    if "whisper" not in sys.modules:
        w = types.ModuleType("whisper")
        wt = types.ModuleType("whisper.tokenizer")
        wt.LANGUAGES = {}
        w.tokenizer = wt
        sys.modules["whisper"] = w
        sys.modules["whisper.tokenizer"] = wt
    # (4) torch 2.10 dropped typing re-exports used by wenet/squeezeformer/conv2d.py:17
    import torch.nn.modules.conv as tconv
    for n in ("Union", "Optional", "Tuple", "List"):
        if not hasattr(tconv, n):
            setattr(tconv, n, getattr(typing, n))
    # (1) py3.12-only multi-line f-strings in model/online_rnnt_model.py: join them, exec as module
    import model  # noqa: F401  (package dir in reference)
    src = open(f"{REF}/model/online_rnnt_model.py", encoding="utf-8").read()
    src = re.sub(r"\{\n\s+", "{", src)
    mod = types.ModuleType("model.online_rnnt_model")
    mod.__file__ = f"{REF}/model/online_rnnt_model.py"
    sys.modules["model.online_rnnt_model"] = mod
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    return mod
