"""Host-side engine: device context, the ``CustomNetwork`` counterpart
(``Network``) and shape-specialised plans, all thin shims over the C-ABI of
``libsup3r_hip.so``.  torch is used ONLY as the device-memory container /
stream owner (``torch.Tensor.data_ptr()`` pointers cross the ABI); no torch op
computes anything on the hot path.

Mirrors, for the hot path, what sup3r gets from ``phygnn.CustomNetwork``
(``.layers``, ``.weights``, ``.save`` / ``.load`` — sup3r/models/abstract.py:
57-111,312-319; base.py:133-214).
"""
import ctypes as C
import os
import pickle

import numpy as np

from . import _lib
from . import spec as S


def _torch():
    import torch
    return torch


class Device:
    """One HIP context per (process, GPU): owns the s3_ctx bound to torch's
    current stream on that device."""

    _cache = {}
    _option_names = None

    def __init__(self, index=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError(
                'sup3r_amd needs an AMD GPU (torch.cuda.is_available() is '
                'False); there is no CPU fallback')
        self.index = index
        self.torch_device = torch.device('cuda', index)
        L = _lib.lib()
        with torch.cuda.device(index):
            stream = torch.cuda.current_stream().cuda_stream
        h = C.c_void_p()
        rc = L.s3_ctx_create(index, C.c_void_p(stream), 0, C.byref(h))
        _lib.check(rc, h, 's3_ctx_create')
        self.ctx = h
        self.rank, self.nranks = 0, 1
        self._set, self.options_key = {}, ()   # options changed since creation
        # a list while a training step is being recorded (captured.py): every
        # buffer handed out then must live as long as the recorded graph
        self._retain = None

    @classmethod
    def get(cls, index=None):
        if index is None:
            index = int(os.environ.get('LOCAL_RANK', 0))
        if index not in cls._cache:
            cls._cache[index] = cls(index)
        return cls._cache[index]

    def sync(self):
        _lib.check(_lib.lib().s3_ctx_sync(self.ctx), self.ctx, 'sync')

    # seconds the host waits for a step whose gradients cross RCCL before it
    # gives the communicator up (``wait``; ``None`` = no deadline)
    comm_timeout_s = 600.0

    def wait(self, timeout_s=None):
        """Bounded host wait for everything enqueued so far — the watchdog
        of the collectives (``s3_comm_wait``): raises ``TimeoutError`` after
        ``timeout_s`` (default ``comm_timeout_s``) with the communicator
        aborted, ``RuntimeError`` on an asynchronous RCCL error."""
        if timeout_s is None:
            timeout_s = self.comm_timeout_s
        ms = -1 if timeout_s is None else int(round(float(timeout_s) * 1e3))
        rc = _lib.lib().s3_comm_wait(self.ctx, ms)
        if rc != 0:
            self.rank, self.nranks = 0, 1      # (the C side fell back too)
        _lib.check(rc, self.ctx, 's3_comm_wait')

    def option_names(self):
        if Device._option_names is None:
            Device._option_names = _lib.option_names()
        return Device._option_names

    def stat(self, name):
        """launch counter of a run-time kernel choice (``_lib.STATS``)"""
        return int(_lib.lib().s3_ctx_stat(self.ctx, _lib.STATS[name]))

    def empty(self, shape):
        torch = _torch()
        t = torch.empty(tuple(int(v) for v in shape), dtype=torch.float32,
                        device=self.torch_device)
        if self._retain is not None:
            self._retain.append(t)
        return t

    def to_device(self, arr):
        """numpy / torch / anything with .numpy() -> contiguous fp32 device
        tensor (batch payloads are duck-typed like the reference does,
        preprocessing/utilities.py:255-257)."""
        torch = _torch()
        if self._retain is not None and not (
                isinstance(arr, torch.Tensor) and arr.is_cuda
                and arr.dtype == torch.float32 and arr.is_contiguous()):
            raise RuntimeError('an upload / conversion inside a recorded '
                               'step: it would not be part of the replay')
        if isinstance(arr, torch.Tensor):
            return arr.to(device=self.torch_device,
                          dtype=torch.float32).contiguous()
        if hasattr(arr, 'numpy') and not isinstance(arr, np.ndarray):
            arr = arr.numpy()
        arr = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
        return torch.from_numpy(arr).to(self.torch_device)

    # -- options: kernel-selection switches of this context (defaults of the
    # plans created from it; include/sup3r_hip.h).  The SUP3R_AMD_<NAME>
    # environment is read once, when the context is created.
    def set_option(self, name, value=1):
        """value None removes the option (default behaviour)"""
        rc = _lib.lib().s3_ctx_set_option(
            self.ctx, name.encode(),
            _lib.OPTION_UNSET if value is None else int(value))
        _lib.check(rc, self.ctx, f's3_ctx_set_option({name})')
        self._set[name] = None if value is None else int(value)
        self.options_key = tuple(sorted(self._set.items()))

    def get_option(self, name):
        v = C.c_int32()
        rc = _lib.lib().s3_ctx_get_option(self.ctx, name.encode(),
                                          C.byref(v))
        if rc < 0:
            raise KeyError(f'unknown option "{name}"')
        return int(v.value) if rc == 1 else None

    def options(self, **values):
        """``with dev.options(NO_PERSIST=1): ...`` — set for the block, the
        previous values restored afterwards"""
        dev = self

        class _Scope:
            def __enter__(self):
                self.old = {k: dev.get_option(k) for k in values}
                for k, v in values.items():
                    dev.set_option(k, v)
                return dev

            def __exit__(self, *exc):
                for k, v in self.old.items():
                    dev.set_option(k, v)
        return _Scope()

    def init_comm(self, rank, nranks, unique_id):
        rc = _lib.lib().s3_comm_init(self.ctx, rank, nranks, unique_id)
        _lib.check(rc, self.ctx, 's3_comm_init')
        self.rank, self.nranks = rank, nranks


def precision_code(precision=None):
    precision = precision or os.environ.get('SUP3R_AMD_PRECISION', 'f32')
    if precision not in _lib.PRECISIONS:
        raise KeyError(f'precision must be one of {list(_lib.PRECISIONS)}')
    return _lib.PRECISIONS[precision]


class PlanHandle:
    """One s3_plan (fixed input shape, precision, training flag)."""

    def __init__(self, net, plan, precision, training, options=None):
        L = _lib.lib()
        self.net, self.plan = net, plan
        self.dev = net.dev
        self.precision, self.training = precision, bool(training)
        nt, nops = len(plan.tensors), len(plan.ops)
        tens = (_lib.TensorDesc * nt)()
        for i, sh in enumerate(plan.tensors):
            for j in range(5):
                tens[i].dims[j] = sh[j]
        ops = (_lib.OpDesc * nops)()
        for i, op in enumerate(plan.ops):
            d = ops[i]
            d.kind = op['kind']
            d.in0, d.in1 = op.get('in0', -1), op.get('in1', -1)
            d.res, d.out = op.get('res', -1), op['out']
            d.w, d.b = op.get('w', -1), op.get('b', -1)
            for j in range(3):
                d.k[j] = op.get('k', [1, 1, 1])[j]
                d.stride[j] = op.get('stride', [1, 1, 1])[j]
                d.lo[j] = op.get('lo', [0, 0, 0])[j]
                d.hi[j] = op.get('hi', [0, 0, 0])[j]
            d.pad_mode = op.get('pad_mode', 0)
            d.act = op.get('act', 0)
            d.alpha = op.get('alpha', 0.0)
            d.d2s = op.get('d2s', 1)
            d.rep = op.get('rep', 1)
            d.bcast_c = op.get('bcast_c', 0)
        self.input_names = list(plan.inputs)
        inp = (C.c_int32 * len(self.input_names))(
            *[plan.inputs[k] for k in self.input_names])
        h = C.c_void_p()
        popt, keep = _lib.plan_options(options)
        rc = L.s3_plan_create_opt(
            self.dev.ctx, net.params, tens, nt, ops, nops, inp,
            len(self.input_names), plan.output, precision, int(training),
            C.byref(popt) if options else None, C.byref(h))
        del keep
        _lib.check(rc, self.dev.ctx, 's3_plan_create')
        self.h = h
        self.out_shape = plan.out_shape
        self.in_shapes = {k: plan.tensors[v] for k, v in plan.inputs.items()}

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                _lib.lib().s3_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _input_ptrs(self, x, exo):
        ptrs = (C.c_void_p * len(self.input_names))()
        keep = []
        for i, name in enumerate(self.input_names):
            t = x if name == 'x' else exo[name]
            want = int(np.prod(self.in_shapes[name]))
            if t.numel() != want:
                raise RuntimeError(
                    f'input "{name}" has {tuple(t.shape)} but the plan '
                    f'expects {self.in_shapes[name]}')
            keep.append(t)
            ptrs[i] = t.data_ptr()
        return ptrs, keep

    def forward(self, x, exo=None, out=None):
        """x / exo: contiguous fp32 device tensors.  Returns a device tensor of
        the keras-view output shape."""
        ptrs, keep = self._input_ptrs(x, exo or {})
        if out is None:
            out = self.dev.empty(self.out_shape)
        rc = _lib.lib().s3_plan_forward(self.h, ptrs, C.c_void_p(out.data_ptr()))
        _lib.check(rc, self.dev.ctx, 's3_plan_forward')
        self._keep = keep
        return out

    @property
    def supports_window(self):
        return bool(_lib.lib().s3_plan_supports_window(self.h))

    def forward_window(self, x, exo, out, lo, n, affine=None):
        """``forward`` whose last conv computes only the window ``[lo, lo +
        n)`` of its output positions, un-normalised by ``affine`` (device
        tensor: scale[C] then shift[C]) and written densely into ``out``
        (``s3_plan_forward_window``)."""
        ptrs, keep = self._input_ptrs(x, exo or {})
        i64x3 = C.c_int64 * 3
        n_c = int(affine.numel() // 2) if affine is not None else 0
        rc = _lib.lib().s3_plan_forward_window(
            self.h, ptrs, C.c_void_p(out.data_ptr()),
            i64x3(*[int(v) for v in lo]), i64x3(*[int(v) for v in n]),
            C.c_void_p(affine.data_ptr()) if affine is not None else None,
            n_c)
        _lib.check(rc, self.dev.ctx, 's3_plan_forward_window')
        self._keep = keep + [affine]
        return out

    def backward(self, d_out, need_dx=False, need_wgrad=True,
                 accumulate_wgrad=False):
        dx = None
        if need_dx:
            dx = self.dev.empty(self.in_shapes['x'])
        rc = _lib.lib().s3_plan_backward(
            self.h, C.c_void_p(d_out.data_ptr()),
            C.c_void_p(dx.data_ptr()) if dx is not None else None,
            int(need_wgrad), int(accumulate_wgrad))
        _lib.check(rc, self.dev.ctx, 's3_plan_backward')
        return dx

    def profile_begin(self, max_forwards):
        """Record HIP events between ops for the next forwards."""
        rc = _lib.lib().s3_plan_profile_begin(self.h, int(max_forwards))
        _lib.check(rc, self.dev.ctx, 's3_plan_profile_begin')

    def profile_end(self):
        """-> (n_forwards_averaged, [mean ms per op])."""
        n = len(self.plan.ops)
        ms = (C.c_float * n)()
        rc = _lib.lib().s3_plan_profile_end(self.h, ms, n)
        if rc < 0:
            _lib.check(rc, self.dev.ctx, 's3_plan_profile_end')
        return rc, [ms[i] for i in range(n)]

    def op_is_mfma(self, i):
        return bool(_lib.lib().s3_plan_op_is_mfma(self.h, i))

    def op_kernel_class(self, i):
        """0 = direct/generic, 1 = MFMA halo tile, 2 = persistent MFMA."""
        return int(_lib.lib().s3_plan_op_is_mfma(self.h, i))

    def op_info(self, i):
        """Kernel selection of op ``i`` (s3_plan_op_info) as a dict with the
        kernel names of ``_lib.FWD_KERNELS`` / ``WGRAD_KERNELS`` /
        ``DGRAD_KERNELS``."""
        n = len(_lib.OPINFO_FIELDS)
        buf = (C.c_int32 * n)()
        rc = _lib.lib().s3_plan_op_info(self.h, int(i), buf, n)
        if rc < 0:
            _lib.check(rc, self.dev.ctx, 's3_plan_op_info')
        d = dict(zip(_lib.OPINFO_FIELDS, [int(v) for v in buf]))
        d['fwd'] = _lib.FWD_KERNELS[d['fwd']]
        d['wgrad'] = _lib.WGRAD_KERNELS[d['wgrad']]
        d['dgrad'] = _lib.DGRAD_KERNELS[d['dgrad']]
        return d

    def tensor_is_bf16(self, tensor_id):
        return int(_lib.lib().s3_plan_tensor_dtype(self.h, int(tensor_id))) == 1

    def tensor(self, tensor_id):
        """fp32 numpy copy of a plan tensor (training plans keep every
        activation; bf16-stored tensors are widened exactly)."""
        shape = self.plan.tensors[tensor_id]
        n = int(np.prod(shape))
        bf16 = self.tensor_is_bf16(tensor_id)
        host = np.empty(n, np.uint16 if bf16 else np.float32)
        rc = _lib.lib().s3_plan_tensor_read(
            self.h, int(tensor_id), host.ctypes.data_as(C.c_void_p),
            host.nbytes)
        if rc < 0:
            _lib.check(rc, self.dev.ctx, 's3_plan_tensor_read')
        if bf16:
            host = (host.astype(np.uint32) << 16).view(np.float32)
        return host.reshape(shape)

    @property
    def workspace_bytes(self):
        return int(_lib.lib().s3_plan_workspace_bytes(self.h))


class Network:
    """Counterpart of ``phygnn.CustomNetwork`` for the hot path: ordered layer
    specs + one device parameter store + cached plans."""

    _global_seed = None   # set by Sup3rGan.seed()

    def __init__(self, hidden_layers, name=None, device=None, precision=None):
        self.name = name
        self.layers = S.parse_layers(hidden_layers)
        self._dev = device
        self.precision = precision
        self.params = None
        self.param_table = None
        self._plans = {}
        self._pending = None   # keras-layout weights set before build
        self._from_file = False
        self._seed = None

    # -- iteration over layers like ``for layer in model.generator``
    def __iter__(self):
        return iter(self.layers)

    def __len__(self):
        return len(self.layers)

    @property
    def dev(self):
        if self._dev is None:
            self._dev = Device.get()
        return self._dev

    @property
    def built(self):
        return self.params is not None

    def build(self, in_shape, seed=None):
        """Create the parameter store for ``in_shape`` (keras lazy build:
        glorot_uniform kernels, zero biases — base.py:394-437)."""
        if self.built:
            return
        plan = S.build_plan(self.layers, in_shape)
        self.param_table = plan.params
        L = _lib.lib()
        n = len(plan.params)
        sizes = (C.c_int64 * max(n, 1))(
            *[int(np.prod(p['shape'])) for p in plan.params])
        h = C.c_void_p()
        rc = L.s3_params_create(self.dev.ctx, n, sizes, C.byref(h))
        _lib.check(rc, self.dev.ctx, 's3_params_create')
        self.params = h
        if self._pending is not None:
            self.set_weights(self._pending)
            self._pending = None
        else:
            if seed is None:
                seed = self._seed if self._seed is not None else \
                    type(self)._global_seed
            rng = np.random.default_rng(seed)
            ws = []
            for p in plan.params:
                if p['kind'] == 'kernel':
                    ws.append(S.glorot_uniform(p['shape'], rng))
                else:
                    ws.append(np.zeros(p['shape'], np.float32))
            self.set_weights(ws)

    def plan(self, in_shape, training=False, precision=None, slot=0,
             options=None):
        """The shape-specialised plan (cached).  ``options``: kernel-selection
        switches of THIS plan on top of the context's (``Device.set_option``),
        e.g. ``{'NO_PERSIST': 1}``; a plan keeps the options it was created
        with."""
        in_shape = tuple(int(v) for v in in_shape)
        prec = precision_code(precision or self.precision)
        okey = tuple(sorted((options or {}).items()))
        # (the context's options in force when the plan is built belong to
        # its identity too)
        ckey = self.dev.options_key
        key = (in_shape, bool(training), prec, slot, okey, ckey)
        if key not in self._plans:
            self.build(in_shape)
            plan = S.build_plan(self.layers, in_shape,
                                param_table=self.param_table)
            self._plans[key] = PlanHandle(self, plan, prec, training,
                                          options=options)
        return self._plans[key]

    def clear_plans(self):
        self._plans = {}
        # recorded steps (captured.py) hold pointers into the plans' arenas
        self.plan_epoch = getattr(self, 'plan_epoch', 0) + 1

    # -- weights, keras layout / keras order
    def _get(self, which):
        L = _lib.lib()
        out = []
        for i, p in enumerate(self.param_table):
            cshape = S.canonical_shape(p['shape'], p['layout'])
            buf = np.empty(cshape, np.float32)
            rc = L.s3_params_get(self.params, which, i,
                                 buf.ctypes.data_as(C.POINTER(C.c_float)))
            _lib.check(rc, self.dev.ctx, 's3_params_get')
            out.append(S.canonical_to_keras(buf, p['layout']))
        return out

    @property
    def weights(self):
        """keras-order weight arrays.  Before the store is built (a network
        loaded from disk that has not run yet) these are the loaded arrays, so
        ``load -> save`` round-trips without a forward pass in between."""
        if not self.built:
            return [np.array(a) for a in self._pending] \
                if self._pending is not None else []
        return self._get(_lib.BUF_W)

    @property
    def grads(self):
        return self._get(_lib.BUF_G)

    def slots(self, which):
        return self._get({'m': _lib.BUF_M, 'v': _lib.BUF_V}[which])

    def set_weights(self, arrays, which=_lib.BUF_W):
        if not self.built:
            self._pending = [np.asarray(a, np.float32) for a in arrays]
            return
        arrays = list(arrays)
        if len(arrays) != len(self.param_table):
            raise RuntimeError(
                f'expected {len(self.param_table)} weight arrays, got '
                f'{len(arrays)}')
        L = _lib.lib()
        for i, (a, p) in enumerate(zip(arrays, self.param_table)):
            a = np.asarray(a, np.float32)
            if tuple(a.shape) != tuple(p['shape']):
                raise RuntimeError(
                    f'weight #{i} has shape {a.shape}, expected {p["shape"]}')
            c = S.keras_to_canonical(a, p['layout'])
            rc = L.s3_params_set(self.params, which, i,
                                 c.ctypes.data_as(C.POINTER(C.c_float)))
            _lib.check(rc, self.dev.ctx, 's3_params_set')

    def mean_abs(self, which, idx):
        v = C.c_float()
        rc = _lib.lib().s3_params_mean_abs(self.params, which, idx,
                                           C.byref(v))
        _lib.check(rc, self.dev.ctx, 's3_params_mean_abs')
        return float(v.value)

    @property
    def weights_version(self):
        """changes whenever the device weights change"""
        return int(_lib.lib().s3_params_version(self.params)) \
            if self.built else -1

    def zero_grad(self):
        _lib.check(_lib.lib().s3_params_zero_grad(self.params), self.dev.ctx,
                   'zero_grad')

    def adam_step(self, lr, beta_1, beta_2, epsilon, t):
        self.optimizer_step(_lib.OPT_ADAM, [lr, beta_1, beta_2, epsilon], t)

    def optimizer_step(self, kind, hyper, t):
        hp = (C.c_double * len(hyper))(*[float(v) for v in hyper])
        rc = _lib.lib().s3_optimizer_step(self.params, int(kind), hp,
                                          len(hyper), int(t))
        _lib.check(rc, self.dev.ctx, 's3_optimizer_step')

    def optimizer_stage(self, kind, hyper, t):
        """first half of ``optimizer_step`` for a recorded step: the scalars
        of step ``t`` go to the device now (outside the graph) ..."""
        hp = (C.c_double * len(hyper))(*[float(v) for v in hyper])
        rc = _lib.lib().s3_optimizer_stage(self.params, int(kind), hp,
                                           len(hyper), int(t))
        _lib.check(rc, self.dev.ctx, 's3_optimizer_stage')

    def optimizer_step_staged(self, kind):
        """... and the update launch reads them there (recordable)"""
        rc = _lib.lib().s3_optimizer_step_staged(self.params, int(kind))
        _lib.check(rc, self.dev.ctx, 's3_optimizer_step_staged')

    def touch(self):
        """the weights changed behind the host's back (a replayed graph
        stepped the optimizer): packed filter images are stale"""
        _lib.check(_lib.lib().s3_params_touch(self.params), self.dev.ctx,
                   's3_params_touch')

    def arm_allreduce(self, bucket_bytes):
        """the next backward pass that writes this store's gradients reduces
        them over the ranks bucket by bucket while it runs (``bucket_bytes``
        < 0 disarms; without a communicator arming is a no-op).  The pass
        consumes the arming — also when it fails."""
        rc = _lib.lib().s3_params_arm_allreduce(self.params,
                                                int(bucket_bytes))
        _lib.check(rc, self.dev.ctx, 's3_params_arm_allreduce')

    def allreduce_grads(self):
        rc = _lib.lib().s3_params_allreduce_grads(self.params)
        _lib.check(rc, self.dev.ctx, 's3_params_allreduce_grads')

    def broadcast(self, which, root=0):
        rc = _lib.lib().s3_params_broadcast(self.params, int(which), int(root))
        _lib.check(rc, self.dev.ctx, 's3_params_broadcast')
        self.clear_graph_state()

    def clear_graph_state(self):
        """weights changed behind the plans' back (broadcast): nothing to do —
        the store's version counter makes every plan re-pack its filter
        images on the next forward"""

    # -- convenience: numpy in / numpy out
    def __call__(self, x, exo=None, training=False, precision=None):
        dev = self.dev
        xd = dev.to_device(x)
        exod = {k: dev.to_device(v) for k, v in (exo or {}).items()}
        ph = self.plan(tuple(xd.shape), training=training,
                       precision=precision)
        return ph.forward(xd, exod)

    # -- persistence (replaces CustomNetwork.save / .load of the phygnn pkl:
    # same role, own schema — phygnn's pickle format is not available here)
    def save(self, fp):
        if not self.built and self._pending is None and self._from_file:
            raise RuntimeError(
                f'network "{self.name}" was created from a weight file but '
                'holds no weights; refusing to overwrite a checkpoint with an '
                'empty one')
        with open(fp, 'wb') as f:
            pickle.dump({'format': 'sup3r_amd.network.v1', 'name': self.name,
                         'hidden_layers': [dict(L.kwargs, **{'class': L.cls})
                                           for L in self.layers],
                         'weights': self.weights}, f)

    @classmethod
    def load(cls, fp, device=None, precision=None):
        hidden, weights, name = read_network_file(fp)
        net = cls(hidden, name=name, device=device, precision=precision)
        net._from_file = True
        if weights:
            net._pending = weights
        return net


def read_network_file(fp):
    """(hidden_layers, weights | None, name) of a saved network.

    Two layouts: this package's ``sup3r_amd.network.v1`` (``Network.save`` and
    ``tools/convert_phygnn_pkl.py``), and the ``model_params`` dict that
    ``phygnn.CustomNetwork.save`` pickles — builtins and numpy arrays only:
    ``hidden_layers`` (the layer list the network was built from) and
    ``weight_dict`` (expanded layer index -> that layer's ``get_weights()``),
    which flattens to the keras-order weight list.  The second layout is
    restated from phygnn 0.0.33's published source and could not be checked
    against a real file here (phygnn is not installable in this image): the
    converter, run where phygnn is, remains the authoritative route, and a
    pickle that needs phygnn / TF classes to unpickle raises a ``TypeError``
    saying so."""
    try:
        with open(fp, 'rb') as f:
            d = pickle.load(f)
    except (ModuleNotFoundError, AttributeError, ImportError) as e:
        raise TypeError(
            f'{fp} needs classes that are not importable here ({e}); convert '
            'it with tools/convert_phygnn_pkl.py on a machine that has '
            'phygnn') from e
    if isinstance(d, dict) and d.get('format') == 'sup3r_amd.network.v1':
        return d['hidden_layers'], (d['weights'] or None), d.get('name')
    if isinstance(d, dict) and 'hidden_layers' in d and \
            ('weight_dict' in d or 'weights' in d):
        if 'weight_dict' in d:
            wd = d['weight_dict']
            keys = sorted(wd, key=lambda k: int(k))
            flat = [np.asarray(a, np.float32) for k in keys
                    for a in (wd[k] or [])]
        else:
            flat = [np.asarray(a, np.float32) for a in d['weights']]
        return d['hidden_layers'], (flat or None), d.get('name')
    raise TypeError(
        f'{fp} is neither a sup3r_amd network file nor a phygnn '
        'CustomNetwork model_params pickle (keys: '
        f'{sorted(d) if isinstance(d, dict) else type(d).__name__})')
