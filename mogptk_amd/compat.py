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
        if dv is not None:
            dv = _num(dv)
            if dv.ndim == 2:                                     # the reference keeps the per-point variances as a dense diagonal MATRIX (gpr/model.py:423)
                dv = np.diagonal(dv).copy()
        inference = _model.Exact(data_variance=dv, jitter=float(g["jitter"]))
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


# ======================================================================================================================================
# Writing a checkpoint the REFERENCE reads (`mogptk.LoadModel`, reference mogptk/model.py:62-74), without the reference installed.
#
# The file must name the reference's classes (`mogptk.models.mosm.MOSM`, `mogptk.gpr.model.Exact`, ...) and carry the attribute dictionaries
# their methods expect after `pickle.load` (no __init__ runs on that side: everything an __init__ would have set has to be in the state --
# the dense identity `eye`, the quadrature nodes of the likelihood, torch.nn.Module's registries).  A small pickler writes the global
# references and object states directly, so nothing is registered in sys.modules and no class of the reference is needed here; tensors are
# written by torch itself.  Scope: Exact and Titsias inference with a Gaussian likelihood and no mean function, the kernels of the six model
# wrappers (and sums / products of them), the Y transformers.  Anything else raises NotImplementedError naming the object.
# ======================================================================================================================================
class _Global:
    """a name in a module of the reference; pickled as a global reference"""

    def __init__(self, module, qualname):
        self.module, self.qualname = module, qualname

    def __call__(self, *a, **k):                 # save_reduce wants a callable
        raise TypeError("placeholder for %s.%s" % (self.module, self.qualname))


class _Obj:
    """an instance of a class of the reference (or of torch) with the given attribute dictionary"""

    def __init__(self, cls, state):
        self.cls, self.state = cls, state


class _Par:
    """a `mogptk.gpr.parameter.Parameter`: rebuilt on the reference's side by Parameter._rebuild (gpr/parameter.py:157-177)"""

    def __init__(self, args):
        self.args = args


class _RefPickler(pickle._Pickler):
    dispatch = dict(pickle._Pickler.dispatch)

    def _save_global_name(self, g):
        self.save(g.module)
        self.save(g.qualname)
        self.write(pickle.STACK_GLOBAL)
        self.memoize(g)

    def _save_obj(self, o):
        self.save(o.cls)
        self.save(())
        self.write(pickle.NEWOBJ)
        self.memoize(o)
        self.save(o.state)
        self.write(pickle.BUILD)

    def _save_par(self, p):
        self.save_reduce(_PARAMETER_REBUILD, p.args, obj=p)

    dispatch[_Global] = _save_global_name
    dispatch[_Obj] = _save_obj
    dispatch[_Par] = _save_par


_PARAMETER_REBUILD = _Global("mogptk.gpr.parameter", "Parameter._rebuild")
_REF_KERNEL_MODULE = {
    "AddKernel": "kernel", "MulKernel": "kernel", "MixtureKernel": "kernel",
    "SpectralKernel": "singleoutput", "SpectralMixtureKernel": "singleoutput", "SquaredExponentialKernel": "singleoutput",
    "IndependentMultiOutputKernel": "multioutput", "MultiOutputSpectralMixtureKernel": "multioutput", "CrossSpectralKernel": "multioutput",
    "LinearModelOfCoregionalizationKernel": "multioutput", "GaussianConvolutionProcessKernel": "multioutput",
    "MultiOutputHarmonizableSpectralKernel": "multioutput", "MultiOutputSpectralKernel": "multioutput",
    "UncoupledMultiOutputSpectralKernel": "multioutput",
}
_REF_WRAPPER_MODULE = {"MOSM": "models.mosm", "SM": "models.sm", "CSM": "models.csm", "SM_LMC": "models.sm_lmc", "CONV": "models.conv",
                       "MOHSM": "models.mohsm", "Model": "model"}
_REF_TRANSFORMER_STATE = {"TransformDetrend": ("degree", "dim", "coef"), "TransformStandard": ("mean", "std"),
                          "TransformNormalize": ("ymin", "ymax"), "TransformLog": ("shift", "mean"), "TransformLinear": ("bias", "slope")}


def _module_state(own, parameters=(), modules=(), first=None):
    """attribute dictionary of a torch.nn.Module in the order the reference's constructors leave it: what they set before
    torch.nn.Module.__init__ (`first`), the registries that __init__ creates (taken from a fresh Module of the installed torch; a reader with
    a newer torch adds the ones it misses in Module.__setstate__), then the class's own attributes"""
    import torch
    st = dict(first or {})
    st.update(torch.nn.Module().__dict__)
    st["_parameters"].update(parameters)
    st["_modules"].update(modules)
    st.update(own)
    return st


class _Exporter:
    def __init__(self):
        import torch
        self.torch = torch
        self.pars = {}                                           # id(our Parameter) -> its _Par (one object per parameter: pegging keeps identity)

    def tensor(self, a):
        return self.torch.tensor(np.array(a, dtype=np.float64), dtype=self.torch.float64)

    def bound(self, b):
        return None if b is None else self.tensor(b)

    def parameter(self, p):
        got = self.pars.get(id(p))
        if got is not None:
            return got
        if p.prior is not None:
            raise NotImplementedError("parameter %s carries a prior; priors are not written" % p._name)
        out = _Par(None)
        self.pars[id(p)] = out
        peg = None if p.pegged_parameter is None else self.parameter(p.pegged_parameter)
        from collections import OrderedDict
        out.args = (self.torch._utils._rebuild_parameter, (self.tensor(p.data), True, OrderedDict()), p._name, self.bound(p.lower),
                    self.bound(p.upper), None, bool(p.train), peg, p.pegged_transform, int(p.num_parameters))
        return out

    def own_parameters(self, holder):
        return [(n, self.parameter(holder.__dict__[n])) for n in holder.__dict__.get("_order", []) if isinstance(holder.__dict__[n], _gpr.Parameter)]

    def module_list(self, items):
        return _Obj(_Global("torch.nn.modules.container", "ModuleList"), _module_state({}, modules=[(str(i), m) for i, m in enumerate(items)]))

    def kernel(self, k):
        name = type(k).__name__
        where = _REF_KERNEL_MODULE.get(name)
        if where is None or type(k) is not getattr(_gpr, name, None):
            raise NotImplementedError("the checkpoint writer does not cover the kernel %s" % name)
        own = {"input_dims": None if name == "IndependentMultiOutputKernel" else k.input_dims, "_active_dims": None, "output_dims": k.output_dims}
        if "twopi" in k.__dict__:
            own["twopi"] = np.float64(k.twopi)
        mods = []
        if isinstance(k.__dict__.get("kernels"), (list, tuple)):
            mods.append(("kernels", self.module_list([self.kernel(s) for s in k.kernels])))
        return _Obj(_Global("mogptk.gpr." + where, name), _module_state(own, self.own_parameters(k), mods))

    def likelihood(self, lik):
        if type(lik) is not _gpr.GaussianLikelihood:
            raise NotImplementedError("the checkpoint writer covers the Gaussian likelihood only, not %s" % type(lik).__name__)
        q = lik.quadrature                                       # reference gpr/likelihood.py:65-81, :90: the nodes every Likelihood object carries
        quad = _Obj(_Global("mogptk.gpr.likelihood", "GaussHermiteQuadrature"),
                    {"t": self.tensor(q.t).reshape(-1, 1), "w": self.tensor(q.w).reshape(-1, 1), "deg": int(q.deg)})
        return _Obj(_Global("mogptk.gpr.likelihood", "GaussianLikelihood"),
                    _module_state({"quadrature": quad, "output_dims": lik.output_dims}, self.own_parameters(lik)))

    def inference(self, g):
        name = type(g).__name__
        if type(g) not in (_gpr.Exact, _gpr.Titsias):
            raise NotImplementedError("the checkpoint writer covers Exact and Titsias inference, not %s" % name)
        if g.mean is not None:
            raise NotImplementedError("mean functions are not written")
        X = np.asarray(g.X, dtype=np.float64)
        own, first = {}, {}
        if name == "Exact":
            dv = g.data_variance
            # the reference adds this attribute to Kff as it is (gpr/model.py:442, :466): it must be the N x N diagonal matrix its own constructor
            # builds with diagflat (:423) -- a vector would broadcast over every row and load without complaint
            first["data_variance"] = None if dv is None else self.torch.diagflat(self.tensor(np.reshape(dv, -1)))
        own.update({"X": self.tensor(X), "y": self.tensor(np.reshape(g.y, (-1, 1))), "mean": None, "jitter": float(g.jitter),
                    "input_dims": int(X.shape[1]), "_compiled_forward": None})
        pars = []
        if name == "Titsias":
            own["eye"] = self.torch.eye(g.Z.data.shape[0], dtype=self.torch.float64)
            pars.append(("Z", self.parameter(g.Z)))
        else:
            own["eye"] = self.torch.eye(X.shape[0], dtype=self.torch.float64)
        own["log_marginal_likelihood_constant"] = np.float64(0.5 * X.shape[0] * np.log(2.0 * np.pi))
        return _Obj(_Global("mogptk.gpr.model", name),
                    _module_state(own, pars, [("kernel", self.kernel(g.kernel)), ("likelihood", self.likelihood(g.likelihood))], first))

    def transformer(self, t):
        name = type(t).__name__
        keys = _REF_TRANSFORMER_STATE.get(name)
        if keys is None or type(t) is not getattr(_transformer, name, None):
            raise NotImplementedError("the checkpoint writer does not cover the transformer %s" % name)
        return _Obj(_Global("mogptk.transformer", name), {k: getattr(t, k) for k in keys})

    def data(self, d):
        D = d.X.shape[1]
        st = {"X": np.array(d.X, dtype=np.float64), "Y": np.array(d.Y, dtype=np.float64),
              "Y_err": None if d.Y_err is None else np.array(d.Y_err, dtype=np.float64),
              "X_pred": None if d.X_pred is None or d.X_pred is d.X else np.array(d.X_pred, dtype=np.float64),
              "mask": np.array(d.mask, dtype=bool), "F": getattr(d, "F", None),
              "X_dtypes": list(getattr(d, "X_dtypes", [np.dtype("float64")] * D)),
              "Y_transformer": _Obj(_Global("mogptk.transformer", "Transformer"),
                                    {"transformers": [self.transformer(t) for t in d.Y_transformer.transformers]}),
              "removed_ranges": getattr(d, "removed_ranges", [[] for _ in range(D)]),
              "X_labels": list(getattr(d, "X_labels", ["X"] * D)), "name": d.name, "Y_label": getattr(d, "Y_label", "Y")}
        return _Obj(_Global("mogptk.data", "Data"), st)

    def model(self, m):
        name = type(m).__name__
        where = _REF_WRAPPER_MODULE.get(name)
        from . import model as _model, wrappers as _wrappers
        if where is None or type(m) is not (getattr(_wrappers, name, None) or _model.Model):
            raise NotImplementedError("the checkpoint writer does not cover the model class %s" % name)
        st = {"name": m.name, "dataset": _Obj(_Global("mogptk.dataset", "DataSet"), {"channels": [self.data(c) for c in m.dataset.channels]}),
              "is_multioutput": m.gpr.kernel.output_dims is not None, "gpr": self.inference(m.gpr), "iters": int(m.iters),
              "times": np.array(m.times, dtype=np.float64), "losses": np.array(m.losses, dtype=np.float64),
              "errors": np.array(m.errors, dtype=np.float64)}
        for key in ("Q", "Rq", "P"):
            if key in m.__dict__:
                st[key] = m.__dict__[key]
        return _Obj(_Global("mogptk." + where, name), st)


def dump_reference_model(model):
    """bytes of a checkpoint of `model` that the reference's `mogptk.LoadModel` reads (see the block comment above for the scope)"""
    try:
        ex = _Exporter()
    except ImportError as e:
        raise ImportError("a reference checkpoint stores torch tensors: torch must be importable to write one") from e
    top = ex.model(model)
    buf = io.BytesIO()
    _RefPickler(buf, protocol=4).dump(top)
    return buf.getvalue()
