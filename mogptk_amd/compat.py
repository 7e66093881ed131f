"""
Reading checkpoints written by the reference.

`mogptk.Model.save()` pickles the WHOLE model object (reference mogptk/model.py:320-336): `mogptk.*` classes, torch modules and tensors,
numpy arrays; `Parameter.__reduce_ex__` / `_rebuild` carry name, bounds, prior, train flag and pegging next to the raw tensor
(reference gpr/parameter.py:157-177) and `gpr.Model.__getstate__` drops the traced forward (gpr/model.py:131-136).  Such a file cannot be
unpickled without the reference package, and unpickled with it it is a torch model.  `load_reference_model` reads it WITHOUT the reference:
every `mogptk.*` class is replaced by a bag that only records its state, torch rebuilds its own tensors (torch must be importable), and the
bags are turned into the objects of this package -- data set (points, masks, prediction inputs, fitted transformers), wrapper class and
kernel structure, inference (Exact, Titsias, Snelson; OpperArchambeau and Hensman with any of the reference's likelihoods), every parameter's raw value / bounds / train flag / pegging in `parameters()` order, and the
training history.  `mogptk_amd.LoadModel` calls it when a file is not one of its own.
"""
import io
import pickle

import numpy as np

from . import dataset as _dataset
from . import gpr as _gpr
from . import transformer as _transformer


class _Bag:
    """stand-in for any `mogptk.*` class: keeps constructor arguments and pickled state"""
    _mod = _cls = None

    def __init__(self, *args, **kwargs):
        self.__dict__["_args"], self.__dict__["_kw"] = args, kwargs

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def state(self):
        return self.__dict__.get("_state", self.__dict__)

    def cls(self):
        return self._cls


class _RefParameter:
    """what `Parameter._rebuild` received (reference gpr/parameter.py:164-177)"""

    def __init__(self, call, args, name, lower, upper, prior, train, pegged_parameter, pegged_transform, num_parameters):
        t = call(*args)
        self.data = np.array(t.detach().cpu().numpy(), dtype=np.float64)
        self.name, self.prior, self.train = name, prior, bool(train)
        self.lower, self.upper = _num(lower), _num(upper)
        self.pegged_parameter, self.pegged_transform, self.num_parameters = pegged_parameter, pegged_transform, num_parameters


def _num(v):
    if v is None:
        return None
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.array(v, dtype=np.float64)


# What a checkpoint written by the reference's Model.save() needs besides its own classes: torch's tensor / parameter rebuilders, the container
# torch.nn.Module state uses, numpy's array rebuilders and a few builtins -- EXACT (module, name) pairs, nothing resolved by prefix and no dotted
# names (pickle protocol 4 resolves "a.b" attribute by attribute: with whole modules allowed, GLOBAL('torch.serialization', 'os.getcwd') or
# torch._utils._import_dotted_name reached anything importable).  A crafted "checkpoint" naming os.system -- which the reference's plain
# pickle.load would run -- raises instead.
_ALLOWED = {
    "torch._utils": {"_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state"},
    "torch._tensor": {"_rebuild_from_type_v2"},
    "torch.nn.modules.container": {"ModuleList"},
    "torch.nn.parameter": {"Parameter"},
    "torch": {"FloatStorage", "DoubleStorage", "LongStorage", "IntStorage", "BoolStorage", "HalfStorage", "ByteStorage", "Size", "device", "dtype",
              "float32", "float64", "int64", "int32", "bool", "Tensor", "UntypedStorage"},
    "numpy": {"dtype", "ndarray", "float64", "float32", "int64", "int32", "bool_"},
    "numpy.core.multiarray": {"_reconstruct", "scalar"},
    "numpy._core.multiarray": {"_reconstruct", "scalar"},
    "numpy.core.numeric": {"_frombuffer"},
    "numpy._core.numeric": {"_frombuffer"},
    "collections": {"OrderedDict"},
    "builtins": {"set", "frozenset", "slice", "complex", "list", "dict", "tuple", "bytearray", "range", "object", "int", "float", "bool", "str", "bytes"},
    "copyreg": {"_reconstructor"},
    "functools": {"partial"},                                        # peg transforms: partial(operator.mul, c) -- whatever it wraps is resolved through this same list
    "_operator": {"mul", "add", "sub", "truediv", "neg", "pow"},
    "datetime": {"datetime", "timedelta", "date"},
    "pandas._libs.tslibs.timestamps": {"_unpickle_timestamp", "Timestamp"},
}


def _load_storage_bytes(b):
    """torch.storage._load_from_bytes is torch.load(bytes) -- a second, unrestricted pickle inside the first.  A storage needs no more than
    torch's weights-only loader."""
    import io
    import torch
    return torch.load(io.BytesIO(b), weights_only=True)


def _refuse(module, name, what, hint=""):
    raise pickle.UnpicklingError("%s names %s.%s, which a mogptk checkpoint has no use for: refused%s" % (what, module, name, hint))


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "mogptk.gpr.parameter" and name == "Parameter._rebuild":
            return _RefParameter
        if module == "mogptk" or module.startswith("mogptk."):
            return type(name.split(".")[-1], (_Bag,), {"_mod": module, "_cls": name.split(".")[-1]})        # a state bag: nothing of the reference is imported or run
        if module == "torch.storage" and name == "_load_from_bytes":
            return _load_storage_bytes
        if "." not in name and name in _ALLOWED.get(module, ()):
            return super().find_class(module, name)
        _refuse(module, name, "reference checkpoint")


# functions of this package a checkpoint of this package may name (everything else it names here must be a CLASS defined in the package: a
# class is only instantiated, a function would be CALLED with arguments the file chooses)
_NATIVE_FUNCTIONS = {("mogptk_amd.gpr.likelihood", "_link_by_name")}


def _in_package(module):
    return module == "mogptk_amd" or module.startswith("mogptk_amd.")


class _NativeUnpickler(pickle.Unpickler):
    """this package's own checkpoints (Model.save pickles the model): classes DEFINED in mogptk_amd, the few of its functions listed above,
    numpy's array rebuilders and the same small set of builtins -- an object merely reachable through one of the package's modules (an imported
    os, subprocess, pickle ...) is not resolved.  A model that carries user code (a mean function, a Kernel / Likelihood / transformer
    subclass of the caller's) names that code's module: `allow=[...]` admits exactly those objects, `LoadModel(..., trusted=True)` is the
    reference's plain pickle.load."""

    def __init__(self, f, allow=()):
        super().__init__(f)
        self._extra = {}
        for obj in allow or ():
            mod, qual = getattr(obj, "__module__", None), getattr(obj, "__qualname__", getattr(obj, "__name__", None))
            if mod is None or qual is None:
                raise TypeError("allow= takes classes and functions (objects with __module__ and __qualname__), got %r" % (obj,))
            self._extra[(mod, qual)] = obj

    def find_class(self, module, name):
        if (module, name) in self._extra:                      # the caller vouched for this object (nested classes come as dotted names)
            return self._extra[(module, name)]
        hint = "; pass it to LoadModel(filename, allow=[...]) if it is yours, or LoadModel(filename, trusted=True) for a file you trust"
        if "." in name:
            _refuse(module, name, "checkpoint", hint)
        if _in_package(module):
            obj = super().find_class(module, name)
            if _in_package(str(getattr(obj, "__module__", ""))) and (isinstance(obj, type) or (module, name) in _NATIVE_FUNCTIONS):
                return obj
            _refuse(module, name, "checkpoint", hint)
        if not module.startswith("torch") and name in _ALLOWED.get(module, ()):
            return super().find_class(module, name)
        _refuse(module, name, "checkpoint", hint)


def load_native_model(raw, allow=(), trusted=False):
    if trusted:
        return pickle.loads(raw)
    return _NativeUnpickler(io.BytesIO(raw), allow).load()


def is_reference_checkpoint(raw):
    """a pickle that names `mogptk.` modules and none of this package's"""
    return b"mogptk_amd" not in raw and (b"mogptk." in raw or b"cmogptk\n" in raw)


# ---- data ----------------------------------------------------------------------------------------------------------------------------
def _convert_transformer(bag):
    st = bag.state()
    name = bag.cls()
    if name == "TransformLinear":
        return _transformer.TransformLinear(bias=st["bias"], slope=st["slope"])
    cls = getattr(_transformer, name, None)
    if cls is None:
        raise NotImplementedError("checkpoint uses the transformer %s, which this package does not have" % name)
    t = cls.__new__(cls)
    t.__dict__.update(st)
    if name == "TransformStandard":
        t.offset, t.scale = t.mean, t.std
    elif name == "TransformNormalize":
        t.scale = 0.5 * (t.ymax - t.ymin)
        t.offset = t.ymin + t.scale
    return t


def _convert_data(bag):
    st = bag.state()
    d = _dataset.Data(np.array(st["X"], dtype=np.float64), np.array(st["Y"], dtype=np.float64), Y_err=st.get("Y_err"), name=st.get("name"))
    d.mask = np.array(st["mask"], dtype=bool)
    if st.get("X_pred") is not None:
        d.X_pred = np.array(st["X_pred"], dtype=np.float64)
    chain = st["Y_transformer"].state()["transformers"]
    d.Y_transformer = _transformer.Transformer([_convert_transformer(t) for t in chain])
    for key in ("F", "X_labels", "Y_label", "removed_ranges", "X_dtypes"):       # kept as found (labels, latent function, removed ranges)
        if key in st:
            setattr(d, key, st[key])
    return d


def _convert_dataset(bag):
    return _dataset.DataSet(*[_convert_data(c) for c in bag.state()["channels"]])


# ---- parameters in torch.nn.Module.parameters() order ----------------------------------------------------------------------------------
def _mstate(mod):
    """state dict of a module: a bag of the reference, or a real torch container (ModuleList) holding bags"""
    return mod.state() if isinstance(mod, _Bag) else mod.__dict__


def _module_parameters(mod, seen=None, out=None):
    if seen is None:
        seen, out = set(), []
    st = _mstate(mod)
    for p in st.get("_parameters", {}).values():
        if p is not None and id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    for sub in st.get("_modules", {}).values():
        if sub is not None:
            _module_parameters(sub, seen, out)
    return out


def _module_children(mod):
    """sub-modules of a container, in order (torch ModuleList keeps them under string indices)"""
    return [m for m in _mstate(mod).get("_modules", {}).values() if m is not None]


# ---- kernels -------------------------------------------------------------------------------------------------------------------------
def _convert_kernel(bag):
    """same class, same structure, default parameter values (they are overwritten afterwards, in order)"""
    name, st = bag.cls(), bag.state()
    params = st.get("_parameters", {})
    cls = getattr(_gpr, name, None)
    if cls is None:
        raise NotImplementedError("checkpoint uses the kernel %s, which this package does not have" % name)
    idims, odims = st.get("input_dims"), st.get("output_dims")
    mods = st.get("_modules", {})
    if name in ("AddKernel", "MulKernel"):
        return cls(*[_convert_kernel(k) for k in _module_children(mods["kernels"])])
    if name == "MixtureKernel":
        subs = _module_children(mods["kernels"])
        k = cls(_convert_kernel(subs[0]), len(subs))
        return k
    if name == "IndependentMultiOutputKernel":
        return cls(*[_convert_kernel(k) for k in _module_children(mods["kernels"])], output_dims=odims)
    if name == "MultiOutputSpectralMixtureKernel":
        return cls(Q=params["weight"].data.shape[1], output_dims=odims, input_dims=idims)
    if name in ("MultiOutputSpectralKernel", "UncoupledMultiOutputSpectralKernel", "MultiOutputHarmonizableSpectralKernel",
                "GaussianConvolutionProcessKernel"):
        return cls(output_dims=odims, input_dims=idims)
    if name == "CrossSpectralKernel":
        return cls(output_dims=odims, input_dims=idims, Rq=params["amplitude"].data.shape[1])
    if name == "LinearModelOfCoregionalizationKernel":
        subs = [_convert_kernel(k) for k in _module_children(mods["kernels"])]
        return cls(*subs, output_dims=odims, input_dims=idims, Rq=params["weight"].data.shape[2])
    if name == "SpectralMixtureKernel":
        return cls(Q=params["magnitude"].data.shape[0], input_dims=idims)
    try:
        return cls(input_dims=idims)                       # single-output kernels: SquaredExponential, Spectral, Matern, ...
    except TypeError:
        raise NotImplementedError("the checkpoint loader does not know how to construct the kernel %s" % name)


def _assign_parameters(ours, theirs):
    if len(ours) != len(theirs):
        raise ValueError("checkpoint has %d parameters, the rebuilt model %d" % (len(theirs), len(ours)))
    index = {id(p): i for i, p in enumerate(theirs)}
    for mine, ref in zip(ours, theirs):
        if mine.data.shape != ref.data.shape:
            raise ValueError("parameter %s: checkpoint shape %s, rebuilt model %s" % (ref.name, ref.data.shape, mine.data.shape))
        if ref.prior is not None:
            raise NotImplementedError("parameter %s carries a prior object of the reference; priors are not converted" % ref.name)
        mine.lower, mine.upper = ref.lower, ref.upper
        mine.transform = type(mine).to_transform(ref.lower, ref.upper)
        mine.data = ref.data.astype(mine.data.dtype).copy()
        mine.train = ref.train
        mine.num_parameters = ref.num_parameters
        if ref.name is not None:
            mine._name = ref.name
        mine.pegged_parameter = mine.pegged_transform = None
    for mine, ref in zip(ours, theirs):                     # pegging (reference gpr/parameter.py:186-201): same link, same transform callable
        if ref.pegged_parameter is not None:
            j = index.get(id(ref.pegged_parameter))
            if j is None:
                raise ValueError("parameter %s is pegged to a parameter outside the model" % ref.name)
            mine.pegged_parameter, mine.pegged_transform = ours[j], ref.pegged_transform


# ---- likelihoods ---------------------------------------------------------------------------------------------------------------------
def _convert_likelihood(bag, variational=False):
    """same class, same link / degrees of freedom / quadrature degree, default parameter values (overwritten afterwards, in order)"""
    name, st = bag.cls(), bag.state()
    cls = getattr(_gpr, name, None)
    if cls is None or not isinstance(cls, type) or not issubclass(cls, _gpr.Likelihood):
        raise NotImplementedError("checkpoint uses the likelihood %s, which this package does not have" % name)
    if name == "MultiOutputLikelihood":
        return cls(*[_convert_likelihood(l, variational) for l in _module_children(st["_modules"]["likelihoods"])])
    kw = {}
    if "link" in st:                                           # pickled by name: the unpickler turned `mogptk.gpr.likelihood.exp` into a bag TYPE
        link = getattr(_gpr, getattr(st["link"], "_cls", None) or getattr(st["link"], "__name__", ""), None)
        if link is None:
            raise NotImplementedError("likelihood %s: the link function of the checkpoint is not one of the reference's own" % name)
        kw["link"] = link
    quad = st.get("quadrature")
    if quad is not None and name not in ("GaussianLikelihood", "BernoulliLikelihood"):
        kw["quadratures"] = int(quad.state()["deg"])
    if name == "StudentTLikelihood":
        kw["dof"] = float(_num(st["dof"]))
    lik = cls(**kw)
    params = st.get("_parameters", {})
    for pname, ref in params.items():                          # shapes: a per-channel Gaussian scale changes output_dims
        if ref is not None and pname == "scale" and name == "GaussianLikelihood" and ref.data.ndim == 1:
            if variational:
                # GaussianLikelihood.variational_expectation takes a scalar scale only: such a model would load and then fail on its first loss()
                raise NotImplementedError("the checkpoint's variational model has a per-channel Gaussian noise scale; its variational expectation "
                                          "is not implemented here (use MultiOutputLikelihood of Gaussian likelihoods)")
            lik = cls(np.ones(ref.data.shape[0]))
    return lik


# ---- the model -----------------------------------------------------------------------------------------------------------------------
def _convert_model(bag):
    from . import model as _model
    from . import wrappers as _wrappers
    st = bag.state()
    dataset = _convert_dataset(st["dataset"])
    g = st["gpr"].state()
    if g.get("mean") is not None:
        raise NotImplementedError("the checkpoint's model has a mean function object of the reference; not converted")
    inference_name = st["gpr"].cls()
    lik = g["_modules"]["likelihood"]
    if lik.cls() != "GaussianLikelihood" and inference_name not in ("SparseHensman", "Hensman", "OpperArchambeau"):
        raise NotImplementedError("likelihood %s with %s inference: only the variational models take a non-Gaussian likelihood" % (lik.cls(), inference_name))
    if inference_name == "Exact":
        dv = g.get("data_variance")
        inference = _model.Exact(data_variance=None if dv is None else _num(dv), jitter=float(g["jitter"]))
    elif inference_name == "Titsias":
        Z = g["_parameters"]["Z"]
        inference = _model.Titsias(inducing_points=np.array(Z.data), jitter=float(g["jitter"]))
    elif inference_name == "Snelson":
        Z = g["_parameters"]["Z"]
        inference = _model.Snelson(inducing_points=np.array(Z.data), jitter=float(g["jitter"]))
    elif inference_name in ("SparseHensman", "Hensman"):
        sparse = bool(g.get("is_sparse", inference_name == "SparseHensman"))
        inference = _model.Hensman(inducing_points=(np.array(g["_parameters"]["Z"].data) if sparse else None), likelihood=_convert_likelihood(lik, True),
                                   jitter=float(g["jitter"]))
    elif inference_name == "OpperArchambeau":
        inference = _model.OpperArchambeau(likelihood=_convert_likelihood(lik, True), jitter=float(g["jitter"]))
    else:
        raise NotImplementedError("inference %s is not part of this package" % inference_name)
    kernel = _convert_kernel(g["_modules"]["kernel"])
    wrapper = getattr(_wrappers, bag.cls(), None)
    m = _model.Model(dataset, kernel, inference=inference, name=st.get("name"))
    if wrapper is not None:                                  # MOSM / SM / CSM / SM_LMC / CONV / MOHSM: same class, same extra attributes
        m.__class__ = wrapper
        for key in ("Q", "Rq", "P"):
            if key in st:
                setattr(m, key, st[key])
    _assign_parameters(list(m.gpr.parameters()), _module_parameters(st["gpr"]))
    m.iters = int(st.get("iters", 0))
    for key in ("times", "losses", "errors"):
        setattr(m, key, np.array(st.get(key, np.zeros(0)), dtype=np.float64))
    return m


def load_reference_model(source):
    """`source`: path of a file written by the reference's `Model.save` (with its '.npy' ending), or its bytes"""
    if isinstance(source, (bytes, bytearray)):
        raw = bytes(source)
    else:
        with open(source, "rb") as f:
            raw = f.read()
    try:
        import torch  # noqa: F401  (the file's tensors are rebuilt by torch itself)
    except ImportError as e:
        raise ImportError("a reference checkpoint stores torch tensors: torch must be importable to read it") from e
    top = _Unpickler(io.BytesIO(raw)).load()
    if not isinstance(top, _Bag) or "gpr" not in top.state() or "dataset" not in top.state():
        raise ValueError("not a model checkpoint of the reference")
    return _convert_model(top)
