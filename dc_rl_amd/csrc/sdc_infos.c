/* _sdc_infos: the `infos` a HARL runner walks every step, as C types (CPython C API, no third-party dependency).
 *
 * What it replaces (reference): `infos` is a tuple[N] of list[3] of dict that every env process builds, pickles and sends
 * through a pipe each step (harl/envs/env_wrappers.py:168-192, :262-273), and that the single-process runner then walks in
 * Python: `infos[i][0].get(key, 0)` for 10 keys per env in the logger (harl/envs/sustaindc/sustaindc_logger.py:87-101) and
 * `"bad_transition" in info[0].keys()` per env in the buffer insert (harl/runners/on_policy_base_runner.py:459-471).
 * At 4096 envs that is ~55 000 lookups per step.  Materialising 4096 x 3 dicts of ~60 keys per step costs more than the
 * runner's own loop, and lazy views written in Python pay an interpreter-level call per lookup (measured: +60 % on the
 * logger loop).  Here the step's info block stays ONE float32 [N, K] array on the host and
 *
 *   InfoSeq   `infos`        sq_item in C: infos[i] -> the env's cached InfoRow
 *   InfoRow   `infos[i]`     the env's per-agent views (a read-only sequence of n_agents entries; a view is created when it is
 *                            first indexed -- the logger only ever touches agent 0's)
 *   InfoView  `infos[i][a]`  a read-only mapping: get / [] / in / keys() in C -- a column of the row becomes a Python float
 *                            only when it is read; per-env constant entries come from a shared dict; everything else
 *                            (derived entries, the `original_*` entries of a finished env) goes back to the Python
 *                            subclass (`_slow_get`, `_full_keys`).
 *
 * The Python side (dc_rl_amd/vec_env.py) subclasses InfoSeq (`LazyInfos`) and hands it a SOURCE object that provides `rows()`
 * (the guarded device -> host copy of the block, called once on first use) and the slow paths `_slow_get`, `_full_keys`,
 * `_has_extra`.  Ownership is a chain without cycles -- InfoSeq -> cached lists -> InfoView -> InfoCore -> source -- so the
 * rows and views need not be garbage-collector-tracked objects.  That matters more than any lookup: 4096 tracked containers
 * per step that survive until the next step march through the collector's generations, and inside a process that has
 * imported torch a full collection is a 35-55 ms pause -- measured as +2.9 ms per runner step at 4096 envs with `infos[i]` a
 * plain list, against 0.1 ms for the walk itself.  Nothing here touches the GPU. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>

/* what the views of one step read from (one per step, shared by the InfoSeq and all of its views) */
typedef struct {
  PyObject_HEAD
  Py_ssize_t n_envs, n_agents;
  PyObject* schema;      /* dict: info key -> column index (int) */
  PyObject* keys_view;   /* the keys of a view without extra entries (a dict_keys object: C-level `in`, iteration order) */
  PyObject* consts;      /* list[N] of dict: per-env constant entries (shared dict objects) */
  PyObject* source;      /* rows(), _slow_get(env, agent, key), _full_keys(env, agent), _has_extra(env, agent) */
  PyObject* rows_obj;    /* the object `rows()` returned (keeps the buffer alive) */
  Py_buffer rows;        /* float32 [N, K], C-contiguous */
  int have_rows;
  int has_extras;        /* some (env, agent) carries extra entries: only then keys() / `in` ask the source */
} InfoCore;

typedef struct {
  PyObject_HEAD
  InfoCore* core;
  PyObject* items;       /* list[N]: cached per-env lists of views (None until first access) */
  PyObject* dict;        /* instance __dict__ (for Python subclasses) */
  PyObject* weaklist;
} InfoSeq;

typedef struct {
  PyObject_HEAD
  InfoCore* core;        /* strong; the core references no view and no InfoSeq: no cycle, so this type is not GC-tracked */
  Py_ssize_t env, agent;
} InfoView;

typedef struct {
  PyObject_VAR_HEAD      /* ob_size = n_agents */
  InfoCore* core;        /* strong; not GC-tracked for the same reason */
  Py_ssize_t env;
  PyObject* views[1];    /* the agents' views, created on first access */
} InfoRow;

static PyTypeObject InfoCore_Type;
static PyTypeObject InfoSeq_Type;
static PyTypeObject InfoView_Type;
static PyTypeObject InfoRow_Type;
static PyObject *str_rows, *str_slow_get, *str_full_keys, *str_has_extra;

/* ------------------------------------------------------------------------------------------------ InfoCore */
static int core_traverse(InfoCore* c, visitproc visit, void* arg) {
  Py_VISIT(c->schema); Py_VISIT(c->keys_view); Py_VISIT(c->consts); Py_VISIT(c->source); Py_VISIT(c->rows_obj);
  return 0;
}
static int core_clear(InfoCore* c) {
  if (c->have_rows) { PyBuffer_Release(&c->rows); c->have_rows = 0; }
  Py_CLEAR(c->schema); Py_CLEAR(c->keys_view); Py_CLEAR(c->consts); Py_CLEAR(c->source); Py_CLEAR(c->rows_obj);
  return 0;
}
static void core_dealloc(InfoCore* c) {
  PyObject_GC_UnTrack(c);
  core_clear(c);
  Py_TYPE(c)->tp_free((PyObject*)c);
}
/* the step's info block on the host: asked of the source once (it checks that the block has not been overwritten) */
static int core_need_rows(InfoCore* c) {
  if (c->have_rows) return 0;
  if (!c->source) { PyErr_SetString(PyExc_RuntimeError, "infos: no source"); return -1; }
  PyObject* r = PyObject_CallMethodNoArgs(c->source, str_rows);
  if (!r) return -1;
  if (PyObject_GetBuffer(r, &c->rows, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { Py_DECREF(r); return -1; }
  if (c->rows.ndim != 2 || c->rows.itemsize != 4 || !c->rows.format || c->rows.format[0] != 'f' || c->rows.shape[0] != c->n_envs) {
    PyBuffer_Release(&c->rows);
    Py_DECREF(r);
    PyErr_SetString(PyExc_TypeError, "infos: rows() must return a C-contiguous float32 array of shape [n_envs, n_columns]");
    return -1;
  }
  c->rows_obj = r;
  c->have_rows = 1;
  return 0;
}
static PyObject* core_call(InfoCore* c, PyObject* name, Py_ssize_t env, Py_ssize_t agent, PyObject* key) {
  if (!c->source) { PyErr_SetString(PyExc_RuntimeError, "infos: no source"); return NULL; }
  PyObject *e = PyLong_FromSsize_t(env), *a = PyLong_FromSsize_t(agent);
  PyObject* r = (e && a) ? PyObject_CallMethodObjArgs(c->source, name, e, a, key, NULL) : NULL;   /* (key may be NULL: end of list) */
  Py_XDECREF(e); Py_XDECREF(a);
  return r;
}
static PyTypeObject InfoCore_Type = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "dc_rl_amd._sdc_infos.InfoCore",
    .tp_basicsize = sizeof(InfoCore),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC,
    .tp_dealloc = (destructor)core_dealloc,
    .tp_traverse = (traverseproc)core_traverse,
    .tp_clear = (inquiry)core_clear,
};

/* ------------------------------------------------------------------------------------------------ InfoSeq */
static int seq_traverse(InfoSeq* s, visitproc visit, void* arg) {
  Py_VISIT(s->core); Py_VISIT(s->items); Py_VISIT(s->dict);
  return 0;
}
static int seq_clear(InfoSeq* s) {
  Py_CLEAR(s->core); Py_CLEAR(s->items); Py_CLEAR(s->dict);
  return 0;
}
static void seq_dealloc(InfoSeq* s) {
  PyObject_GC_UnTrack(s);
  if (s->weaklist) PyObject_ClearWeakRefs((PyObject*)s);
  seq_clear(s);
  Py_TYPE(s)->tp_free((PyObject*)s);
}
static int seq_init(InfoSeq* s, PyObject* args, PyObject* kw) {
  static char* names[] = {"n_envs", "n_agents", "schema", "keys_view", "consts", "source", "has_extras", NULL};
  Py_ssize_t n, k;
  int has_extras = 0;
  PyObject *schema, *kv, *consts, *source;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "nnO!OO!O|p", names, &n, &k, &PyDict_Type, &schema, &kv, &PyList_Type, &consts, &source,
                                   &has_extras))
    return -1;
  if (n < 0 || k < 1 || PyList_GET_SIZE(consts) != n) {
    PyErr_SetString(PyExc_ValueError, "InfoSeq: consts must be a list with one dict per env");
    return -1;
  }
  InfoCore* c = PyObject_GC_New(InfoCore, &InfoCore_Type);
  if (!c) return -1;
  c->n_envs = n; c->n_agents = k;
  Py_INCREF(schema); c->schema = schema;
  Py_INCREF(kv); c->keys_view = kv;
  Py_INCREF(consts); c->consts = consts;
  Py_INCREF(source); c->source = source;
  c->rows_obj = NULL; c->have_rows = 0; c->has_extras = has_extras;
  PyObject_GC_Track(c);
  Py_XSETREF(s->core, c);     /* (a second __init__ replaces the first one's state; the instance __dict__ stays) */
  Py_CLEAR(s->items);         /* the row cache is made on first access: a step whose infos nobody reads pays for none of it */
  return 0;
}
static Py_ssize_t seq_length(InfoSeq* s) { return s->core ? s->core->n_envs : 0; }
static PyObject* seq_item(InfoSeq* s, Py_ssize_t i) {
  if (!s->core) { PyErr_SetString(PyExc_RuntimeError, "InfoSeq.__init__ was not called"); return NULL; }
  if (i < 0 || i >= s->core->n_envs) { PyErr_SetString(PyExc_IndexError, "infos index out of range"); return NULL; }
  if (!s->items) {
    const Py_ssize_t n = s->core->n_envs;
    s->items = PyList_New(n);
    if (!s->items) return NULL;
    for (Py_ssize_t j = 0; j < n; j++) { Py_INCREF(Py_None); PyList_SET_ITEM(s->items, j, Py_None); }
  }
  PyObject* it = PyList_GET_ITEM(s->items, i);
  if (it == Py_None) {
    /* one entry per TRAINED agent, in the reference's order (harlsustaindc_env.py:118-123) */
    const Py_ssize_t k = s->core->n_agents;
    InfoRow* r = PyObject_NewVar(InfoRow, &InfoRow_Type, k);
    if (!r) return NULL;
    Py_INCREF(s->core); r->core = s->core; r->env = i;
    for (Py_ssize_t a = 0; a < k; a++) r->views[a] = NULL;
    it = (PyObject*)r;
    PyList_SetItem(s->items, i, it);   /* steals `it`, drops None */
  }
  Py_INCREF(it);
  return it;
}
static PyObject* seq_subscript(InfoSeq* s, PyObject* key) {
  const Py_ssize_t len = seq_length(s);
  if (PyIndex_Check(key)) {
    Py_ssize_t i = PyNumber_AsSsize_t(key, PyExc_IndexError);
    if (i == -1 && PyErr_Occurred()) return NULL;
    if (i < 0) i += len;
    return seq_item(s, i);
  }
  if (PySlice_Check(key)) {
    Py_ssize_t start, stop, step;
    if (PySlice_Unpack(key, &start, &stop, &step) < 0) return NULL;
    Py_ssize_t n = PySlice_AdjustIndices(len, &start, &stop, step);
    PyObject* out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t j = 0; j < n; j++) {
      PyObject* it = seq_item(s, start + j * step);
      if (!it) { Py_DECREF(out); return NULL; }
      PyList_SET_ITEM(out, j, it);
    }
    return out;
  }
  PyErr_SetString(PyExc_TypeError, "infos indices must be integers or slices");
  return NULL;
}
static PySequenceMethods seq_as_sequence = {.sq_length = (lenfunc)seq_length, .sq_item = (ssizeargfunc)seq_item};
static PyMappingMethods seq_as_mapping = {.mp_length = (lenfunc)seq_length, .mp_subscript = (binaryfunc)seq_subscript};
static PyTypeObject InfoSeq_Type = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "dc_rl_amd._sdc_infos.InfoSeq",
    .tp_basicsize = sizeof(InfoSeq),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE | Py_TPFLAGS_HAVE_GC,
    .tp_doc = "infos of one step: sequence[N] of list[n_agents] of InfoView over one float32 [N, K] block",
    .tp_new = PyType_GenericNew,
    .tp_init = (initproc)seq_init,
    .tp_dealloc = (destructor)seq_dealloc,
    .tp_traverse = (traverseproc)seq_traverse,
    .tp_clear = (inquiry)seq_clear,
    .tp_as_sequence = &seq_as_sequence,
    .tp_as_mapping = &seq_as_mapping,
    .tp_dictoffset = offsetof(InfoSeq, dict),
    .tp_weaklistoffset = offsetof(InfoSeq, weaklist),
};

/* ------------------------------------------------------------------------------------------------ InfoView */
static void view_dealloc(InfoView* v) {
  Py_CLEAR(v->core);
  Py_TYPE(v)->tp_free((PyObject*)v);
}
/* new reference, or NULL with KeyError (or another error) set */
static PyObject* view_lookup(InfoView* v, PyObject* key) {
  InfoCore* c = v->core;
  PyObject* idx = PyDict_GetItemWithError(c->schema, key);   /* borrowed */
  if (idx) {
    if (core_need_rows(c)) return NULL;
    const Py_ssize_t j = PyLong_AsSsize_t(idx);
    if (j < 0 || j >= c->rows.shape[1]) { PyErr_SetString(PyExc_IndexError, "info column out of range"); return NULL; }
    return PyFloat_FromDouble((double)((const float*)c->rows.buf)[v->env * c->rows.shape[1] + j]);
  }
  if (PyErr_Occurred()) return NULL;
  PyObject* k = PyDict_GetItemWithError(PyList_GET_ITEM(c->consts, v->env), key);
  if (k) { Py_INCREF(k); return k; }
  if (PyErr_Occurred()) return NULL;
  return core_call(c, str_slow_get, v->env, v->agent, key);
}
static PyObject* view_subscript(InfoView* v, PyObject* key) { return view_lookup(v, key); }
static PyObject* view_get(InfoView* v, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs < 1 || nargs > 2) { PyErr_SetString(PyExc_TypeError, "get(key[, default])"); return NULL; }
  PyObject* r = view_lookup(v, args[0]);
  if (r) return r;
  if (!PyErr_ExceptionMatches(PyExc_KeyError)) return NULL;
  PyErr_Clear();
  PyObject* d = nargs == 2 ? args[1] : Py_None;
  Py_INCREF(d);
  return d;
}
/* keys(): the shared key view unless this (env, agent) carries extra entries (a finished env's `original_*`) */
static PyObject* view_keys(InfoView* v, PyObject* Py_UNUSED(ignored)) {
  InfoCore* c = v->core;
  if (c->has_extras) {
    PyObject* r = core_call(c, str_has_extra, v->env, v->agent, NULL);
    if (!r) return NULL;
    const int ex = PyObject_IsTrue(r);
    Py_DECREF(r);
    if (ex < 0) return NULL;
    if (ex) return core_call(c, str_full_keys, v->env, v->agent, NULL);
  }
  Py_INCREF(c->keys_view);
  return c->keys_view;
}
static PyObject* view_iter(InfoView* v) {
  PyObject* k = view_keys(v, NULL);
  if (!k) return NULL;
  PyObject* it = PyObject_GetIter(k);
  Py_DECREF(k);
  return it;
}
static Py_ssize_t view_length(InfoView* v) {
  PyObject* k = view_keys(v, NULL);
  if (!k) return -1;
  const Py_ssize_t n = PyObject_Length(k);
  Py_DECREF(k);
  return n;
}
static int view_contains(InfoView* v, PyObject* key) {
  InfoCore* c = v->core;
  int r = PyDict_Contains(c->schema, key);
  if (r != 0) return r;
  r = PyDict_Contains(PyList_GET_ITEM(c->consts, v->env), key);
  if (r != 0) return r;
  PyObject* k = view_keys(v, NULL);
  if (!k) return -1;
  r = PySequence_Contains(k, key);
  Py_DECREF(k);
  return r;
}
static PyObject* view_items(InfoView* v, PyObject* Py_UNUSED(ignored)) {
  PyObject* k = view_keys(v, NULL);
  if (!k) return NULL;
  PyObject* it = PyObject_GetIter(k);
  Py_DECREF(k);
  if (!it) return NULL;
  PyObject* out = PyList_New(0);
  PyObject* key;
  while (out && (key = PyIter_Next(it))) {
    PyObject* val = view_lookup(v, key);
    PyObject* t = val ? PyTuple_Pack(2, key, val) : NULL;
    Py_XDECREF(val); Py_DECREF(key);
    if (!t || PyList_Append(out, t) != 0) { Py_XDECREF(t); Py_CLEAR(out); break; }
    Py_DECREF(t);
  }
  Py_DECREF(it);
  if (out && PyErr_Occurred()) Py_CLEAR(out);
  return out;
}
static PyObject* view_values(InfoView* v, PyObject* Py_UNUSED(ignored)) {
  PyObject* items = view_items(v, NULL);
  if (!items) return NULL;
  const Py_ssize_t n = PyList_GET_SIZE(items);
  PyObject* out = PyList_New(n);
  for (Py_ssize_t i = 0; out && i < n; i++) {
    PyObject* val = PyTuple_GET_ITEM(PyList_GET_ITEM(items, i), 1);
    Py_INCREF(val);
    PyList_SET_ITEM(out, i, val);
  }
  Py_DECREF(items);
  return out;
}
static PyObject* view_repr(InfoView* v) {
  return PyUnicode_FromFormat("<InfoView env %zd agent %zd>", v->env, v->agent);
}
static PyMethodDef view_methods[] = {
    {"get", (PyCFunction)(void (*)(void))view_get, METH_FASTCALL, "get(key[, default])"},
    {"keys", (PyCFunction)view_keys, METH_NOARGS, NULL},
    {"items", (PyCFunction)view_items, METH_NOARGS, NULL},
    {"values", (PyCFunction)view_values, METH_NOARGS, NULL},
    {NULL}};
static PyMemberDef view_members[] = {{"env", T_PYSSIZET, offsetof(InfoView, env), READONLY, NULL},
                                     {"agent", T_PYSSIZET, offsetof(InfoView, agent), READONLY, NULL},
                                     {NULL}};
static PyMappingMethods view_as_mapping = {.mp_length = (lenfunc)view_length, .mp_subscript = (binaryfunc)view_subscript};
static PySequenceMethods view_as_sequence = {.sq_contains = (objobjproc)view_contains};
static PyTypeObject InfoView_Type = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "dc_rl_amd._sdc_infos.InfoView",
    .tp_basicsize = sizeof(InfoView),
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_doc = "one agent's info dict of one env: a read-only mapping over the step's info row",
    .tp_dealloc = (destructor)view_dealloc,
    .tp_as_mapping = &view_as_mapping,
    .tp_as_sequence = &view_as_sequence,
    .tp_iter = (getiterfunc)view_iter,
    .tp_methods = view_methods,
    .tp_members = view_members,
    .tp_repr = (reprfunc)view_repr,
};

/* ------------------------------------------------------------------------------------------------ InfoRow */
static void row_dealloc(InfoRow* r) {
  for (Py_ssize_t a = 0; a < Py_SIZE(r); a++) Py_CLEAR(r->views[a]);
  Py_CLEAR(r->core);
  Py_TYPE(r)->tp_free((PyObject*)r);
}
static Py_ssize_t row_length(InfoRow* r) { return Py_SIZE(r); }
static PyObject* row_item(InfoRow* r, Py_ssize_t a) {
  if (a < 0 || a >= Py_SIZE(r)) { PyErr_SetString(PyExc_IndexError, "agent index out of range"); return NULL; }
  PyObject* v = r->views[a];
  if (!v) {
    InfoView* nv = PyObject_New(InfoView, &InfoView_Type);
    if (!nv) return NULL;
    Py_INCREF(r->core); nv->core = r->core; nv->env = r->env; nv->agent = a;
    v = r->views[a] = (PyObject*)nv;
  }
  Py_INCREF(v);
  return v;
}
static PyObject* row_subscript(InfoRow* r, PyObject* key) {
  const Py_ssize_t len = Py_SIZE(r);
  if (PyIndex_Check(key)) {
    Py_ssize_t i = PyNumber_AsSsize_t(key, PyExc_IndexError);
    if (i == -1 && PyErr_Occurred()) return NULL;
    if (i < 0) i += len;
    return row_item(r, i);
  }
  if (PySlice_Check(key)) {
    Py_ssize_t start, stop, step;
    if (PySlice_Unpack(key, &start, &stop, &step) < 0) return NULL;
    Py_ssize_t n = PySlice_AdjustIndices(len, &start, &stop, step);
    PyObject* out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t j = 0; j < n; j++) {
      PyObject* it = row_item(r, start + j * step);
      if (!it) { Py_DECREF(out); return NULL; }
      PyList_SET_ITEM(out, j, it);
    }
    return out;
  }
  PyErr_SetString(PyExc_TypeError, "indices must be integers or slices");
  return NULL;
}
static PyObject* row_repr(InfoRow* r) { return PyUnicode_FromFormat("<infos of env %zd: %zd agents>", r->env, Py_SIZE(r)); }
static PySequenceMethods row_as_sequence = {.sq_length = (lenfunc)row_length, .sq_item = (ssizeargfunc)row_item};
static PyMappingMethods row_as_mapping = {.mp_length = (lenfunc)row_length, .mp_subscript = (binaryfunc)row_subscript};
static PyTypeObject InfoRow_Type = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "dc_rl_amd._sdc_infos.InfoRow",
    .tp_basicsize = offsetof(InfoRow, views),
    .tp_itemsize = sizeof(PyObject*),
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_doc = "infos[i]: the per-agent info mappings of one env (list[n_agents] of dict in the reference)",
    .tp_dealloc = (destructor)row_dealloc,
    .tp_as_sequence = &row_as_sequence,
    .tp_as_mapping = &row_as_mapping,
    .tp_repr = (reprfunc)row_repr,
};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_sdc_infos", "C types behind SustainDCVecEnv's `infos`", -1, NULL};
PyMODINIT_FUNC PyInit__sdc_infos(void) {
  if (PyType_Ready(&InfoCore_Type) < 0 || PyType_Ready(&InfoSeq_Type) < 0 || PyType_Ready(&InfoView_Type) < 0 ||
      PyType_Ready(&InfoRow_Type) < 0)
    return NULL;
  PyObject* m = PyModule_Create(&moddef);
  if (!m) return NULL;
  str_rows = PyUnicode_InternFromString("rows");
  str_slow_get = PyUnicode_InternFromString("_slow_get");
  str_full_keys = PyUnicode_InternFromString("_full_keys");
  str_has_extra = PyUnicode_InternFromString("_has_extra");
  Py_INCREF(&InfoSeq_Type);
  Py_INCREF(&InfoView_Type);
  Py_INCREF(&InfoRow_Type);
  if (PyModule_AddObject(m, "InfoRow", (PyObject*)&InfoRow_Type) < 0 || PyModule_AddObject(m, "InfoSeq", (PyObject*)&InfoSeq_Type) < 0 || PyModule_AddObject(m, "InfoView", (PyObject*)&InfoView_Type) < 0 ||
      PyModule_AddIntConstant(m, "VERSION", 4) < 0) {
    Py_DECREF(m);
    return NULL;
  }
  return m;
}
