"""Loading the reference's checkpoints into the product classes.

The reference keeps its models as WHOLE-MODULE pickles: ``torch.save(model, open(path, 'wb'))`` when the validation loss improves
and once more at the end (train_generative.py:198-213), and it loads the click model the same way
(``torch.load(open(args.resp_path, 'rb'))``, train_generative.py:259; pretrain_env.py saves it likewise).  Such a pickle names its
class by the reference's module path (``models.pivotcvae.UserPivotCVAE``, ``env.response_model.UserResponseModel_MLP`` ...), which
does not exist next to this package - and the reference's own ``__dict__`` layout is not the product's.  So:

  1. the pickle is read with an ``Unpickler`` whose ``find_class`` maps every class of the reference's model / environment modules
     to an inert shell ``nn.Module`` (no reference code is imported or needed) and resolves ONLY an allowlist of other globals
     (torch's tensor / parameter rebuild functions, ``nn.Module`` layer classes, ``OrderedDict``, a few builtins): anything else
     raises ``UnpicklingError`` instead of being imported and called - ``torch.load(weights_only=False)`` on its own would run it;
  2. the shell holds what the reference object held: hyper-parameters as attributes, parameters through ``state_dict()``;
  3. the PRODUCT class of the same name is built through its normal constructor from those hyper-parameters and takes the
     shell's ``state_dict`` (key names are identical by design: SURVEY.md 8b) - bit for bit, the frozen tables included.

``state_dict`` files (``torch.save(model.state_dict())``) need none of this: ``model.load_state_dict(torch.load(path))``.
"""
import pickle

import torch
from torch import nn

# modules of the reference whose classes may appear in a checkpoint
_REF_MODULES = ("models.pivotcvae", "models.listcvae", "models.cvae", "env.response_model")


class ReferenceShell(nn.Module):
    """Receives the ``__dict__`` of a pickled reference module.  Never runs: it only holds attributes and parameters."""
    ref_module = ref_name = None

    def forward(self, *a, **k):   # pragma: no cover
        raise RuntimeError(f"{self.ref_module}.{self.ref_name} is a loaded reference checkpoint, not a model: "
                           "pivotcvae_amd.checkpoint.from_reference_module() builds the product model from it")


_shells = {}


def _shell(module, name):
    key = (module, name)
    if key not in _shells:
        _shells[key] = type(name, (ReferenceShell,), {"ref_module": module, "ref_name": name, "__module__": __name__})
    return _shells[key]


# Everything ELSE a module pickle may name: an allowlist.  torch.load(weights_only=False) runs a full pickle machine - a "checkpoint"
# may name any importable callable and have it called on load - so the unpickler here refuses every global that a whole-module pickle
# of these model classes does not need.  (What it still trusts: torch's own tensor / parameter rebuild functions and nn.Module
# classes.  A checkpoint from an untrusted source is better converted to a state_dict by its owner.)
_ALLOWED = {
    "collections": {"OrderedDict", "defaultdict"},
    "builtins": {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "complex", "bytes", "bytearray", "slice",
                 "range", "object", "str"},
    "torch._utils": {"_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state"},
    "torch": {"Size", "device", "dtype", "Tensor", "FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage",
              "IntStorage", "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "float32", "float64", "float16", "bfloat16",
              "int64", "int32", "int16", "int8", "uint8", "bool"},
    # NOT torch.storage._load_from_bytes: it is torch.load(BytesIO(b), weights_only=False) on the STANDARD unpickler - a pickle that
    # REDUCEs it over an inner pickle would run that inner payload (regression test: the nested payload is refused).
    # torch.save writes storages through persistent ids; only a tensor pickled by plain pickle.dumps would need it.
    "torch.storage": {"UntypedStorage", "TypedStorage"},
    "torch.nn.parameter": {"Parameter", "Buffer"},
    "torch._tensor": {"_rebuild_from_type_v2"},
}
_ALLOWED["__builtin__"] = _ALLOWED["builtins"]   # (Python-2 style name pickle still emits for set)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module in _REF_MODULES:
            return _shell(module, name)
        if name in _ALLOWED.get(module, ()):
            return super().find_class(module, name)
        if module.startswith("torch.nn.modules."):   # the layers a model holds: nn.Module subclasses only, nothing callable besides
            obj = super().find_class(module, name)
            if isinstance(obj, type) and issubclass(obj, nn.Module):
                return obj
        raise pickle.UnpicklingError(f"reference checkpoint names {module}.{name}: not something a pickled PivotCVAE / click-model "
                                     "module needs - refused (pivotcvae_amd.checkpoint loads model pickles, not arbitrary ones)")


class _PickleModule:
    """what torch.load(pickle_module=...) needs: an Unpickler class and load()"""
    __name__ = "pivotcvae_amd.checkpoint"
    Unpickler = _Unpickler

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()


def read_reference_pickle(path_or_file):
    """-> the unpickled object with every reference class replaced by a ReferenceShell (tensors on the CPU)"""
    return torch.load(path_or_file, map_location="cpu", pickle_module=_PickleModule, weights_only=False)


def _embedding(w):
    return nn.Embedding.from_pretrained(w.detach().clone().float(), freeze=True)


def _struct_of(sd, prefix):
    """layer widths of the nn.Linear stack ``prefix_1 .. prefix_n`` in a state_dict"""
    dims, i = [], 1
    while f"{prefix}_{i}.weight" in sd:
        w = sd[f"{prefix}_{i}.weight"]
        dims = [w.shape[1], w.shape[0]] if not dims else dims + [w.shape[0]]
        i += 1
    return dims


def from_reference_module(shell, device):
    """ReferenceShell (or anything with the reference's attributes + state_dict) -> the product model on ``device``."""
    from .env import response_model as env
    from .models import listcvae, pivotcvae
    name = shell.ref_name if isinstance(shell, ReferenceShell) else type(shell).__name__
    sd = {k: v.detach().cpu() for k, v in shell.state_dict().items()}
    g = lambda attr, default=None: getattr(shell, attr, default)   # noqa: E731
    if hasattr(pivotcvae, name) and isinstance(getattr(pivotcvae, name), type) and issubclass(getattr(pivotcvae, name), pivotcvae.UserPivotCVAE):
        cls = getattr(pivotcvae, name)
        no_user = bool(g("noUser", "userEmbed.weight" not in sd))
        m = cls(_embedding(sd["docEmbed.weight"]), None if no_user else _embedding(sd["userEmbed.weight"]), int(g("slate_size")),
                int(g("feature_size")), int(g("latent_size")), int(g("condition_size")), list(g("encoderStruct")),
                list(g("psmStruct")), list(g("scmStruct")), list(g("priorStruct")), no_user, device)
        m.candidateFlag = bool(g("candidateFlag", False))
    elif name == "UserListCVAEWithPrior":
        no_user = bool(g("noUser", "userEmbed.weight" not in sd))
        m = listcvae.UserListCVAEWithPrior(_embedding(sd["docEmbed.weight"]), None if no_user else _embedding(sd["userEmbed.weight"]),
                                           int(g("slate_size")), int(g("feature_size")), int(g("latent_size")),
                                           int(g("condition_size")), list(g("encoderStruct")), list(g("decoderStruct")),
                                           list(g("priorStruct")), no_user, device)
        m.candidateFlag = bool(g("candidateFlag", False))
    elif name == "UserResponseModel_MLP":
        m = env.UserResponseModel_MLP(int(g("maxItemId")), int(g("maxUserId")), int(g("featureSize")), int(g("slateSize")),
                                      _struct_of(sd, "mlp"), device, bool(g("noUser", False)))
    elif name in ("URM", "URM_P", "URM_P_MR"):
        args = [int(g("maxItemId")), int(g("maxUserId")), int(g("slateSize")), int(g("featureSize")), device, bool(g("noUser", False))]
        if name != "URM":
            args += [float(g("p_bias_max")), float(g("p_bias_min"))]
        if name == "URM_P_MR":
            args += [float(g("mrFactor"))]
        m = getattr(env, name)(*args)
        for attr in ("posBias", "posDependentBias"):   # plain tensors of the reference object, not parameters (:277-285)
            if g(attr) is not None:
                setattr(m, attr, torch.as_tensor(g(attr)).detach().float().to(device))
    else:
        raise ValueError(f"no product class for the reference's {getattr(shell, 'ref_module', '?')}.{name}")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"reference checkpoint of {name}: state_dict keys differ (missing {missing}, unexpected {unexpected})")
    return m.to(device)


def load_reference_pickle(path_or_file, device="cuda:0"):
    """``torch.save(model, ...)`` of the reference (train_generative.py:198-213; the click model of :259) -> the product model of
    the same class on ``device``: same hyper-parameters, same parameters and frozen tables bit for bit."""
    obj = read_reference_pickle(path_or_file)
    if not isinstance(obj, ReferenceShell):
        raise TypeError(f"{path_or_file}: not a pickled reference model (got {type(obj).__name__}); a state_dict loads with "
                        "model.load_state_dict(torch.load(path))")
    return from_reference_module(obj, device)
