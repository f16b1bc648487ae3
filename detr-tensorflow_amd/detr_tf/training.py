"""Training / evaluation loop of the drop-in API (reference detr_tf/training.py:9-87).

`run_train_step`, `run_val_step`, `fit` and `eval` keep the reference's signatures, return values, console lines and
side effects (`config.global_step`).  Underneath, a step is a fixed sequence of HIP kernel launches with no host
synchronisation and no allocation, so `fit` can record it ONCE as a hipGraph (`GraphedTrainStep`) and replay it for
every further batch of the same shape: whatever changes between steps -- the batch, the dropout step seed, the Adam step
sizes -- lives in device memory that is refreshed before each replay.
"""
import os
import time

import torch

from . import _hip as hip
from .loss import hungarian_matching as _matching
from .loss.loss import get_losses, log_from_losses
from .optimizers import GROUPS, aggregate_grad_and_apply, gather_gradient

PRINT_EVERY_TRAIN = 100         # training.py:57
PRINT_EVERY_VAL = 10            # training.py:80


def _gradient_aggregate(config):
    if config.target_batch is not None:
        return max(1, int(config.target_batch // config.batch_size))
    return 1


def run_train_step(model, images, t_bbox, t_class, optimizers, config):
    """training.py:9-25: forward (training=True), set loss / gradient_aggregate, gradients."""
    gradient_aggregate = _gradient_aggregate(config)
    optimizers["_engine"] = model.engine
    m_outputs = model(images, training=True)
    _matching.before_assign = model.engine.pregen_dropmasks_at_matcher
    try:
        total_loss, log = get_losses(m_outputs, t_bbox, t_class, config)
    finally:
        _matching.before_assign = None
    total_loss = total_loss / gradient_aggregate
    gradient_steps = gather_gradient(model, optimizers, total_loss, m_outputs, config, log,
                                     loss_scale=1.0 / gradient_aggregate)
    return m_outputs, total_loss, log, gradient_steps


def run_val_step(model, images, t_bbox, t_class, config):
    """training.py:28-32."""
    m_outputs = model(images, training=False)
    total_loss, log = get_losses(m_outputs, t_bbox, t_class, config)
    return m_outputs, total_loss, log


def train_step(model, images, t_bbox, t_class, optimizers, config, epoch_step):
    """What `fit` does for one batch (training.py:46-54): gradients, then accumulate / apply per group."""
    m_outputs, total_loss, log, gradient_steps = run_train_step(model, images, t_bbox, t_class, optimizers, config)
    model.engine.phase("optimizer")
    for name in gradient_steps:
        aggregate_grad_and_apply(name, optimizers, gradient_steps[name]["gradients"], epoch_step, config)
    model.engine.phase("end")
    return m_outputs, total_loss, log


class _Segments:
    """A step recorded as consecutive hipGraphs with an EAGER action between them (the data-parallel collectives stay
    outside the graphs: RCCL launches on its own stream as soon as the segment that finalises a gradient bucket is queued)."""

    def __init__(self):
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs, self.actions = [], []
        self._cur = None
        self._pad = torch.zeros(4, dtype=torch.int32, device="cuda")

    def begin(self):
        self._cur = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread's calls are checked against the capture.  In the default (global) mode any thread's
        # unsafe call aborts the capture -- and the process group's watchdog thread polls its work events with
        # hipEventQuery at arbitrary times (seen once as a SIGABRT out of a helper thread during the recording of a step
        # with a live RCCL group: tests/test_gpu_dp.py::test_bench_forced_single_rank_rccl_path)
        self._cur.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        hip.zero_(self._pad)               # a segment may record nothing else (after the last gradient bucket): never an empty graph

    def cut(self, action):
        self._cur.capture_end()
        self.graphs.append(self._cur)
        self.actions.append(action)
        self.begin()

    def end(self):
        self._cur.capture_end()
        self.graphs.append(self._cur)
        self.actions.append(None)
        self._cur = None

    def replay(self):
        for g, act in zip(self.graphs, self.actions):
            g.replay()
            if act is not None:
                act()


class GraphedTrainStep:
    """train_step() for a fixed batch shape, with the launch path chosen by `launch`:

      "graph"  the step is recorded once as hipGraph(s) and replayed;
      "eager"  one launch call per kernel on two HIP streams (train_step);
      "auto"   (default of `fit`) both: after the eager warm-up step and the recording pass, PROBE_STEPS steps of each are
               timed (they are ordinary training steps) and the faster path is kept -- which one wins depends on the host:
               replay is one call per step but hipGraph serialises most of the second stream's work, the eager step keeps
               the overlap but needs a host that issues ~600 launches in < 19 ms.  Data-parallel ranks agree through a MAX
               all-reduce of the two timings.  `choice` / `probe` say what was decided (bench.py prints them).

    The first `eager_steps` calls run eagerly (they also allocate every buffer of the static memory plan); the next call
    records the launch sequence -- forward, set loss, backward into one graph (cut at the data-parallel exchange points
    when training on several GPUs), clip + Adam of the three groups into another -- and from then on a replayed step is:
    copy the batch into the static input buffers, write the new dropout seed / Adam step sizes to device memory, replay.
    A new batch shape (or set of trained groups), and any re-allocation of an engine buffer since the recording (a forward
    of another shape in between: engine.buf_generation), drop the recording: one eager step, then a new one.  Gradient
    accumulation and `config.check_matching` (a host synchronisation inside the loss) run eagerly.
    Every call returns FRESH loss / log tensors (clones of the static ones), like the eager path and the reference."""

    PROBE_STEPS = 3

    def __init__(self, model, optimizers, config, eager_steps=1, launch="graph"):
        assert launch in ("graph", "eager", "auto")
        self.model, self.optimizers, self.config = model, optimizers, config
        self.eager_steps = max(1, int(eager_steps))
        self.launch = launch
        self.calls = 0
        self.key = None
        self.step_graph = self.apply_graph = None
        self.static = None
        self.result = None
        self._gen = -1
        self.choice = "eager" if launch == "eager" else "graph"       # the path steady-state steps take
        self.probe = None                                              # {"graph_ms", "eager_ms"} once "auto" has decided (once per stepper)
        self._probe_t = {}
        self._probe_dirty = False
        self._probe_skipped = False
        self.total_calls = 0                                           # never reset: the probe schedule of "auto" runs on it

    def _signature(self, images, t_bbox, t_class):
        c = self.config
        return (tuple(images.shape), tuple(t_bbox.shape), bool(c.train_backbone), bool(c.train_transformers),
                bool(c.train_nlayers), int(c.background_class), self.model.engine.compute, float(self.model.engine.dropout_p),
                bool(getattr(c, "check_matching", False)))

    def _capture(self, images, t_bbox, t_class):
        model, opts, cfg = self.model, self.optimizers, self.config
        eng, dev = model.engine, model.device
        st_img = torch.empty(tuple(images.shape), dtype=torch.float32, device=dev)
        st_tb = torch.empty(tuple(t_bbox.shape), dtype=torch.float32, device=dev)
        st_tc = torch.empty((t_bbox.shape[0], t_bbox.shape[1]), dtype=torch.int64, device=dev)
        self.static = (st_img, st_tb, st_tc)
        self._load_inputs(images, t_bbox, t_class)
        dp = model.dp
        seg = _Segments()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        eng._graph_replay = True
        eng.bump_weights_version()          # the derived weight copies are rebuilt inside the recorded step, as in every eager step
        prev_hooks = None
        try:
            with torch.cuda.stream(side):
                if dp is not None and dp.active:      # collectives run eagerly between graph segments
                    prev_hooks = (dp.on_bucket, dp.reduce_sums, dp.finish)
                    dp.on_bucket = lambda i, f=prev_hooks[0]: seg.cut(lambda: f(i))
                    dp.reduce_sums = lambda sums, f=prev_hooks[1]: seg.cut(lambda: f(sums))
                    dp.finish = lambda: None
                seg.begin()
                m_outputs, total_loss, log, gradient_steps = run_train_step(model, st_img, st_tb, st_tc, opts, cfg)
                seg.end()
                if prev_hooks is not None:
                    dp.on_bucket, dp.reduce_sums, dp.finish = prev_hooks
                    prev_hooks = None
                app = _Segments()
                app.pool = seg.pool
                app.begin()
                store = opts["_store"]
                for name in GROUPS:
                    if getattr(cfg, f"train_{name}"):
                        opts[f"{name}_gradients"] = store.grad
                        opts[f"{name}_optimizer"].apply_gradients(store.grad, hyper_ready=True)
                app.end()
        finally:
            if prev_hooks is not None:
                dp.on_bucket, dp.reduce_sums, dp.finish = prev_hooks
            eng._graph_replay = False
        torch.cuda.current_stream().wait_stream(side)
        self.step_graph, self.apply_graph = seg, app
        self.result = (m_outputs, m_outputs.set_loss)
        self._gen = eng.buf_generation

    def _load_inputs(self, images, t_bbox, t_class):
        st_img, st_tb, st_tc = self.static
        if not (torch.is_tensor(images) and images.data_ptr() == st_img.data_ptr()):
            st_img.copy_(torch.as_tensor(images), non_blocking=True)
        if not (torch.is_tensor(t_bbox) and t_bbox.data_ptr() == st_tb.data_ptr()):
            st_tb.copy_(torch.as_tensor(t_bbox), non_blocking=True)
        tc = torch.as_tensor(t_class)
        if not (tc.is_cuda and tc.data_ptr() == st_tc.data_ptr()):
            st_tc.copy_(tc.reshape(st_tc.shape), non_blocking=True)

    def _replay(self, images, t_bbox, t_class):
        cfg, model = self.config, self.model
        if self.step_graph is None:
            self._capture(images, t_bbox, t_class)
        else:
            self._load_inputs(images, t_bbox, t_class)
        eng = model.engine
        eng._graph_replay = True
        try:
            if eng.dropout_p > 0.0:
                eng.advance_dropout_step()                         # new masks: the kernels read the seed from device memory
            for name in GROUPS:
                if getattr(cfg, f"train_{name}"):
                    self.optimizers[f"{name}_optimizer"].set_step_hyper()
            self.step_graph.replay()
            if model.dp is not None:
                model.dp.finish()
            self.apply_graph.replay()
        finally:
            eng._graph_replay = False
        eng.bump_weights_version()                                 # the parameters moved: derived copies are stale for eager passes
        m_outputs, sl = self.result
        # fresh tensors per step (two small device copies): a caller that keeps per-step losses must not end up with N
        # aliases of the static tensors the graph writes
        log = log_from_losses(sl.losses.clone())
        total_loss = sl.total.clone()[0]
        for name in GROUPS:
            log[f"{name}_lr"] = self.optimizers[f"{name}_optimizer"].learning_rate()
        return m_outputs, total_loss, log

    def _decide(self):
        """Called ONCE per stepper, by every data-parallel rank at the same call count (the probe schedule counts every call of
        the stepper and is never restarted by a rank-local re-recording): the MAX all-reduce below therefore cannot interleave
        with another rank's gradient-bucket collectives.  A rank whose probe window was disturbed (re-recording inside it)
        contributes +inf and the ranks fall back to the eager path together."""
        n = self.PROBE_STEPS
        tg, te = self._probe_t.get("graph_s"), self._probe_t.get("eager_s")
        tg = float("inf") if tg is None else tg
        te = float("inf") if te is None else te
        dp = self.model.dp
        if dp is not None and getattr(dp, "active", False):
            import torch.distributed as dist
            big = 1.0e30
            tt = torch.tensor([min(tg, big), min(te, big)], dtype=torch.float64, device=self.model.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=getattr(dp, "group", None))
            tg, te = (float(v) for v in tt.tolist())
            tg = float("inf") if tg >= big else tg
            te = float("inf") if te >= big else te
        # the graph path is only taken on a COMPARISON it won: a window that was spoiled on either side (no time) falls back to eager
        self.choice = "graph" if (tg != float("inf") and te != float("inf") and tg <= te) else "eager"
        ms = lambda t: None if t == float("inf") else round(t / n * 1e3, 3)
        self.probe = {"graph_ms": ms(tg), "eager_ms": ms(te)}

    def __call__(self, images, t_bbox, t_class, epoch_step):
        cfg, model = self.config, self.model
        eng = model.engine
        self.total_calls += 1
        key = self._signature(images, t_bbox, t_class)
        stale = self.step_graph is not None and self._gen != eng.buf_generation
        if key != self.key or stale:        # new batch shape / trained groups / precision, or the recorded buffers are gone
            self.key, self.calls = key, 0
            self.step_graph = self.apply_graph = self.result = None
            self._probe_dirty = True        # (a probe window this falls into does not count; the DECISION, once taken, stands)
            if self.probe is None:
                self.choice = "eager" if self.launch == "eager" else "graph"
        self.calls += 1
        eager_only = (self.launch == "eager" or _gradient_aggregate(cfg) > 1 or bool(getattr(cfg, "check_matching", False)))
        if eager_only:
            # an eager-only call that lands inside the probe schedule of an "auto" stepper spoils the window it falls into:
            # the window's first step may never set its start time, its last step would time a partial window
            if self.launch == "auto" and self.probe is None:
                self._probe_dirty = True
                self._probe_skipped = True
                if self.total_calls - self.eager_steps - 1 == 2 * self.PROBE_STEPS:
                    # the schedule's last call: the decision is due NOW on every rank (it is a collective under data parallelism) --
                    # with the eager window spoiled it is "eager", explicitly, and a valid graph time stays in the record
                    self._probe_t.pop("eager_s", None)
                    self._decide()
            return train_step(model, images, t_bbox, t_class, self.optimizers, cfg, epoch_step)
        warm = self.calls <= self.eager_steps       # first sighting of a shape: eager (allocates the static memory plan)
        if self.launch != "auto" or self.probe is not None:
            if self.choice == "graph" and not warm:
                return self._replay(images, t_bbox, t_class)
            return train_step(model, images, t_bbox, t_class, self.optimizers, cfg, epoch_step)
        # ---- "auto", still probing.  The schedule counts EVERY call of this stepper (total_calls), so that all data-parallel
        #      ranks reach the decision -- the only collective of the stepper -- at the same call: eager_steps warm-up calls, one
        #      recording call, PROBE_STEPS timed replays, PROBE_STEPS timed eager steps (all ordinary training steps)
        i = self.total_calls - self.eager_steps - 1          # 0 = the recording pass
        n = self.PROBE_STEPS
        if i < 0:
            return train_step(model, images, t_bbox, t_class, self.optimizers, cfg, epoch_step)
        if i == 0:
            self._probe_dirty = False
            return train_step(model, images, t_bbox, t_class, self.optimizers, cfg, epoch_step) if warm else \
                self._replay(images, t_bbox, t_class)
        if i > 2 * n:                        # (the window was spent in eager-only mode: nothing measured, no collective anywhere)
            self.choice, self.probe = "eager", {"graph_ms": None, "eager_ms": None}
            return train_step(model, images, t_bbox, t_class, self.optimizers, cfg, epoch_step)
        phase = "graph" if i <= n else "eager"
        first, last = (i - 1) % n == 0, (i - 1) % n == n - 1
        if first:
            torch.cuda.synchronize()
            self._probe_dirty = warm or (phase == "graph" and self.step_graph is None)     # a (re-)recording inside the window
            self._probe_skipped = False
            self._probe_t[phase] = time.perf_counter()
        out = self._replay(images, t_bbox, t_class) if (phase == "graph" and not warm) else \
            train_step(model, images, t_bbox, t_class, self.optimizers, cfg, epoch_step)
        if last:
            torch.cuda.synchronize()
            if not self._probe_dirty and not getattr(self, "_probe_skipped", False) and phase in self._probe_t:
                self._probe_t[phase + "_s"] = time.perf_counter() - self._probe_t[phase]
            if phase == "eager":
                self._decide()
        return out

    @property
    def settle_calls(self):
        """Calls after which the launch path is final (bench.py warms up at least this long)."""
        return self.eager_steps + 1 + (2 * self.PROBE_STEPS if self.launch == "auto" else 0)


def _launch_mode(config):
    """How `fit` launches a step: config.launch ("auto" | "graph" | "eager") if set; else the legacy switches config.use_graph /
    env DETR_HIP_GRAPH (0 = eager, 1 = graph); else "auto"."""
    mode = getattr(config, "launch", None)
    if mode in ("auto", "graph", "eager"):
        return mode
    flag = getattr(config, "use_graph", None)
    if flag is None and os.environ.get("DETR_HIP_GRAPH") is not None:
        flag = os.environ["DETR_HIP_GRAPH"] != "0"
    if flag is None:
        return "auto"
    return "graph" if flag else "eager"


def _console(prefix, log, elapsed):
    return (f"{prefix}, \t ce: [{float(log['label_cost']):.2f}] \t giou : [{float(log['giou_loss']):.2f}] \t "
            f"l1 : [{float(log['l1_loss']):.2f}] \t time : [{elapsed:.2f}]")


class _Stopwatch:
    """The reference's `t` bookkeeping: 0.00 at the first report; `fit` then measures since the PREVIOUS report
    (training.py:58-63 resets t), `eval` since the FIRST one (training.py:81-83 never resets it)."""

    def __init__(self, reset):
        self.t, self.reset = None, reset

    def lap(self):
        if self.t is None:
            self.t = time.time()
        elapsed = time.time() - self.t
        if self.reset:
            self.t = time.time()
        return elapsed


def fit(model, train_dt, optimizers, config, epoch_nb, class_names):
    """Train the model for one epoch (training.py:35-65); same console line every 100 steps, `config.global_step`
    advanced once per batch.  Launch path: `config.launch` = "auto" (default: GraphedTrainStep probes hipGraph replay against the
    eager two-stream step during the first steps and keeps the faster), "graph" or "eager" (legacy: config.use_graph /
    DETR_HIP_GRAPH=0|1)."""
    mode = _launch_mode(config)
    stepper = optimizers.get("_graphed_step")
    if stepper is None or stepper.launch != mode or stepper.model is not model:
        stepper = optimizers["_graphed_step"] = GraphedTrainStep(model, optimizers, config, launch=mode)
    watch = _Stopwatch(reset=True)
    for epoch_step, (images, t_bbox, t_class) in enumerate(train_dt):
        m_outputs, total_loss, log = stepper(images, t_bbox, t_class, epoch_step)
        if epoch_step % PRINT_EVERY_TRAIN == 0:
            print(_console(f"Epoch: [{epoch_nb}], \t Step: [{epoch_step}]", log, watch.lap()))
        config.global_step += 1


def eval(model, valid_dt, config, class_name, evaluation_step=200):
    """Evaluate on the validation set (training.py:68-87): forward with training=False + set loss for at most
    `evaluation_step` batches, console line every 10."""
    watch = _Stopwatch(reset=False)
    for val_step, (images, t_bbox, t_class) in enumerate(valid_dt):
        m_outputs, total_loss, log = run_val_step(model, images, t_bbox, t_class, config)
        if val_step % PRINT_EVERY_VAL == 0:
            print(_console(f"Validation step: [{val_step}]", log, watch.lap()))
        if val_step + 1 >= evaluation_step:
            break
