"""Kernel-selection options on the host side (`tln_options` of include/tln.h).

The native library holds no process-wide mutable state: every test / measurement switch is a field of a `tln_options`
struct that is stored in a handle (lattice, frame program) or passed with an operator call.  This module is the host
layer's way to say which options are in force: a context manager that keeps ONE struct per host thread,

    with options(v2_min_m=1):            # every product of this block's calls on the large-M kernel
        model(lattice, positions, values)

which `Lattice`, `engine.FrameProgram` and `ops` hand to the library explicitly (tln_lattice_set_options,
tln_program_set_options, tln_gather_gemm_opt ...).  Host threads started by `streams.SequenceStreams` /
`pipeline.FramePipeline` inherit the options of the thread that calls their `run()`.  Outside any `with options(...)`
block nothing is passed and the library's own choices apply.
"""
import contextlib
import ctypes as C
import itertools
import threading

from . import _lib

__all__ = ["options", "push", "set", "pop", "current", "current_ref", "generation", "inherit"]

_tls = threading.local()
_gen = itertools.count(1)


def current():
    """the _lib.Options in force on this host thread, or None"""
    return getattr(_tls, "opt", None)


def generation():
    """changes whenever the options in force on this thread change (0: none): handles cache what they applied"""
    return getattr(_tls, "gen", 0)


def current_ref():
    """what the *_opt entry points take: a ctypes reference to the struct in force, or None (= the library's defaults)"""
    o = current()
    return C.byref(o) if o is not None else None


def make(**fields):
    """a struct with the library's defaults (tln_options_init), then the options in force, then `fields`"""
    o = _lib.Options()
    _lib.lib().tln_options_init(C.byref(o))
    cur = current()
    if cur is not None:
        C.memmove(C.byref(o), C.byref(cur), C.sizeof(o))
    names = {n for n, _ in _lib.Options._fields_}
    for k, v in fields.items():
        if k not in names:
            raise TypeError("tln_options has no field %r (fields: %s)" % (k, ", ".join(sorted(names))))
        setattr(o, k, v)
    return o


def push(**fields):
    """explicit form of `with options(...)` for try / finally code: the options in force + `fields` until pop()"""
    new = make(**fields)              # (raises on an unknown field before anything is pushed)
    stack = getattr(_tls, "stack", None)
    if stack is None:
        stack = _tls.stack = []
    stack.append((current(), generation()))
    _tls.opt, _tls.gen = new, next(_gen)
    return _tls.opt


def set(**fields):
    """changes fields of the options pushed last (a new generation: handles re-apply them)"""
    if not getattr(_tls, "stack", None):
        raise RuntimeError("options.set() outside options.push() / with options(...)")
    _tls.opt, _tls.gen = make(**fields), next(_gen)
    return _tls.opt


def pop():
    _tls.opt, _tls.gen = _tls.stack.pop()


def reset():
    """drops whatever this host thread pushed (tests: a failed test must not leave its options to the next one)"""
    _tls.stack, _tls.opt, _tls.gen = [], None, 0


@contextlib.contextmanager
def options(**fields):
    """kernel-selection options for everything this host thread issues inside the block (fields of tln_options)"""
    push(**fields)
    try:
        yield _tls.opt
    finally:
        pop()


@contextlib.contextmanager
def inherit(opt, gen):
    """a worker thread takes over the (struct, generation) its caller captured with current() / generation()"""
    prev, prev_gen = current(), generation()
    _tls.opt, _tls.gen = opt, gen
    try:
        yield
    finally:
        _tls.opt, _tls.gen = prev, prev_gen
