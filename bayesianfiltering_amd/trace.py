"""Python callables -> device source.

The reference hands plain Python functions ``f(x, q, u)``, ``h(x, r, u)``, ``log_prob(x, y, u)`` to every filter
(gaussfiltax/models.py:46-49, 73-84) and lets JAX trace them.  A Python callable cannot run inside a HIP kernel, but a
function written with NumPy operations can be *recorded*: it is called once with arrays of symbolic scalars (NumPy
dispatches ``np.sin(a)`` on an object array to ``a[i].sin()``, ``A @ x`` to the elements' ``*`` and ``+``), every operation
lands on a tape with common subexpressions shared, and the tape is written out as the ``template <class T>`` source that
``nonlinearities.user_dynamics / user_emission / user_log_prob`` take -- which the engine compiles at run time (hiprtc) with
``T = float`` for values and ``T = dual number`` for the Jacobians ``jacfwd`` would give.

What can be recorded: arithmetic, ``@`` / ``np.dot`` / ``np.sum`` / indexing / ``np.array([...])`` / ``np.concatenate``,
the ufuncs sin cos tan exp log sqrt tanh arctan arctan2 abs square power(x, number), comparisons combined with
:func:`where` (``np.where`` asks the condition for its truth value, which a symbol does not have; compare single elements with
the operators, whole arrays with :data:`greater` / :data:`less` / ...).  Data-dependent Python
control flow (``if x[0] > 0:``) cannot -- it raises ``TraceError`` with this explanation.  Closure constants become
literals of the source (float32): a function is compiled once per distinct set of constants (1-2 s by hiprtc, cached on disk by
source), so a sweep over thousands of parameter values is better served by ``nonlinearities.user_dynamics(source, theta=...)``,
whose ``theta`` is a run-time argument.
"""
import numpy as np

F32 = np.float32


class TraceError(TypeError):
    pass


def _lit(v):
    v = F32(v)
    if np.isnan(v):
        return "__builtin_nanf(\"\")"
    if np.isinf(v):
        return ("-" if v < 0 else "") + "__builtin_inff()"
    return float(v).hex() + "f"          # exact, C++17 hexadecimal floating literal


class _Tape:
    def __init__(self):
        self.stmts, self.cse, self.n = [], {}, 0

    def emit(self, expr, boolean=False):
        ref = self.cse.get(expr)
        if ref is None:
            ref = f"{'c' if boolean else 't'}{self.n}"
            self.n += 1
            self.stmts.append(f"  const {'bool' if boolean else 'T'} {ref} = {expr};")
            self.cse[expr] = ref
        return ref


class Sym:
    """One symbolic fp32 scalar on a tape."""
    __array_priority__ = 1000.0
    __slots__ = ("tape", "ref")

    def __init__(self, tape, ref):
        self.tape, self.ref = tape, ref

    # ---- helpers
    def _wrap(self, other):
        if isinstance(other, Sym):
            if other.tape is not self.tape:
                raise TraceError("symbols of two different traces met")
            return other.ref, False
        if isinstance(other, (bool, np.bool_)):
            return _lit(float(other)), True
        if isinstance(other, (int, float, np.integer, np.floating)):
            return _lit(other), True
        return None, False

    def _bin(self, other, op, swap=False):
        ref, is_num = self._wrap(other)
        if ref is None:
            return NotImplemented
        if is_num:   # exact identities with literal 0 / 1 (0 * x -> 0 assumes a finite x: dense matrices with structural zeros)
            v = float(other)
            if op == "*" and v == 1.0:
                return self
            if op == "*" and v == 0.0:
                return 0.0
            if op == "+" and v == 0.0:
                return self
            if op == "-" and v == 0.0 and not swap:
                return self
            if op == "/" and v == 1.0 and not swap:
                return self
            if op == "/" and v == 0.0 and swap:
                return 0.0
        a, b = (ref, self.ref) if swap else (self.ref, ref)
        return Sym(self.tape, self.tape.emit(f"{a} {op} {b}"))

    def _call(self, name, *args):
        refs = []
        for a in args:
            r, _ = self._wrap(a)
            if r is None:
                raise TraceError(f"{name}: unsupported operand {type(a).__name__}")
            refs.append(r)
        return Sym(self.tape, self.tape.emit(f"{name}({', '.join([self.ref] + refs)})"))

    # ---- arithmetic
    def __add__(self, o): return self._bin(o, "+")
    def __radd__(self, o): return self._bin(o, "+", True)
    def __sub__(self, o): return self._bin(o, "-")
    def __rsub__(self, o): return self._bin(o, "-", True)
    def __mul__(self, o): return self._bin(o, "*")
    def __rmul__(self, o): return self._bin(o, "*", True)
    def __truediv__(self, o): return self._bin(o, "/")
    def __rtruediv__(self, o): return self._bin(o, "/", True)
    def __neg__(self): return Sym(self.tape, self.tape.emit(f"-{self.ref}"))
    def __pos__(self): return self
    def __abs__(self): return self._call("abs")

    def __pow__(self, p):
        if isinstance(p, Sym):
            return (self.log() * p).exp()
        p = float(p)
        if p == int(p) and 0 <= int(p) <= 8:
            n = int(p)
            if n == 0:
                return Sym(self.tape, self.tape.emit(_lit(1.0)))
            out = self
            for _ in range(n - 1):
                out = out * self
            return out
        if p == 0.5:
            return self.sqrt()
        if p == int(p) and -8 <= int(p) < 0:
            return 1.0 / self.__pow__(-int(p))
        return Sym(self.tape, self.tape.emit(f"pow({self.ref}, {_lit(p)})"))

    def __rpow__(self, base):          # number ** symbol
        return (self * float(np.log(float(base)))).exp()

    # ---- comparisons (for where)
    def _cmp(self, o, op):
        ref, _ = self._wrap(o)
        if ref is None:
            return NotImplemented
        return SymBool(self.tape, self.tape.emit(f"{self.ref} {op} {ref}", boolean=True))

    def __lt__(self, o): return self._cmp(o, "<")
    def __le__(self, o): return self._cmp(o, "<=")
    def __gt__(self, o): return self._cmp(o, ">")
    def __ge__(self, o): return self._cmp(o, ">=")
    def __eq__(self, o): return self._cmp(o, "==")     # (so that `if u[0] == 1:` stops the recording instead of silently taking
    def __ne__(self, o): return self._cmp(o, "!=")     #  the identity comparison's False branch)
    __hash__ = None

    def __bool__(self):
        raise TraceError("the truth value of a traced quantity was asked for (a Python `if` / `and` / np.where on the state): "
                         "data-dependent control flow cannot be recorded -- use bayesianfiltering_amd.trace.where(cond, a, b)")

    def __float__(self):
        raise TraceError("a traced quantity was converted to a Python float: the function leaves NumPy-recordable operations here")

    __int__ = __index__ = __float__

    # ---- ufunc methods (NumPy calls these for object arrays)
    def sin(self): return self._call("sin")
    def cos(self): return self._call("cos")
    def tan(self): return self._call("tan")
    def exp(self): return self._call("exp")
    def log(self): return self._call("log")
    def sqrt(self): return self._call("sqrt")
    def tanh(self): return self._call("tanh")
    def arctan(self): return self._call("atan")
    def arctan2(self, other): return self._call("atan2", other)
    def absolute(self): return self._call("abs")
    fabs = absolute
    def square(self): return self * self
    def reciprocal(self): return 1.0 / self
    def negative(self): return -self
    def conjugate(self): return self
    def log1p(self): return (1.0 + self).log()
    def expm1(self): return self.exp() - 1.0

    def __repr__(self):
        return f"Sym({self.ref})"


class SymBool:
    __slots__ = ("tape", "ref")

    def __init__(self, tape, ref):
        self.tape, self.ref = tape, ref

    def _comb(self, o, op):
        if isinstance(o, SymBool):
            return SymBool(self.tape, self.tape.emit(f"{self.ref} {op} {o.ref}", boolean=True))
        if isinstance(o, (bool, np.bool_)):
            return SymBool(self.tape, self.tape.emit(f"{self.ref} {op} {'true' if o else 'false'}", boolean=True))
        return NotImplemented

    def __and__(self, o): return self._comb(o, "&&")
    __rand__ = __and__
    def __or__(self, o): return self._comb(o, "||")
    __ror__ = __or__
    def __invert__(self): return SymBool(self.tape, self.tape.emit(f"!{self.ref}", boolean=True))

    def __bool__(self):
        raise TraceError("the truth value of a traced comparison was asked for (a Python `if` / np.where on the state): use "
                         "bayesianfiltering_amd.trace.where(cond, a, b)")


def where(cond, a, b):
    """``np.where`` for traced conditions (element-wise on arrays); plain NumPy for plain arguments."""
    ca = np.asarray(cond, dtype=object) if isinstance(cond, (SymBool, list, tuple, np.ndarray)) else cond
    if not (isinstance(ca, np.ndarray) and ca.dtype == object) and not isinstance(cond, SymBool):
        return np.where(cond, a, b)
    ca, aa, ba = np.broadcast_arrays(np.asarray(ca, dtype=object), np.asarray(a, dtype=object), np.asarray(b, dtype=object))
    out = np.empty(ca.shape, dtype=object)
    for idx in np.ndindex(ca.shape):
        c, x, y = ca[idx], aa[idx], ba[idx]
        if isinstance(c, SymBool):
            xr = x.ref if isinstance(x, Sym) else _lit(x)
            yr = y.ref if isinstance(y, Sym) else _lit(y)
            out[idx] = Sym(c.tape, c.tape.emit(f"({c.ref} ? T({xr}) : T({yr}))"))
        else:
            out[idx] = x if c else y
    return out if out.shape else out[()]


# comparisons of ARRAYS of traced quantities (the operators on object arrays ask each result for its truth value; on single
# elements -- x[0] > 1.0 -- the operators themselves work)
greater = np.frompyfunc(lambda a, b: a > b, 2, 1)
less = np.frompyfunc(lambda a, b: a < b, 2, 1)
greater_equal = np.frompyfunc(lambda a, b: a >= b, 2, 1)
less_equal = np.frompyfunc(lambda a, b: a <= b, 2, 1)


def _symbols(tape, name, n):
    return np.array([Sym(tape, f"{name}[{i}]") for i in range(n)], dtype=object)


def _record(fn, args, what):
    try:
        res = fn(*args)
    except TraceError:
        raise
    except Exception as e:       # a jnp.* call, a float() of the state, ...
        raise TraceError(f"{what}: the Python function could not be recorded as NumPy operations on symbols "
                         f"({type(e).__name__}: {e}); write it with numpy operations (bayesianfiltering_amd.trace), or give it as "
                         f"source with nonlinearities.user_*") from e
    return res


def _finish(tape, outs, header, ret_scalar=False):
    lines = [header + " {"] + tape.stmts
    if ret_scalar:
        o = outs[0]
        lines.append(f"  return T({o.ref if isinstance(o, Sym) else _lit(o)});")
    else:
        for i, o in enumerate(outs):
            lines.append(f"  out[{i}] = T({o.ref if isinstance(o, Sym) else _lit(o)});")
    lines.append("}")
    return "\n".join(lines) + "\n"


def _flat(res, what):
    arr = np.asarray(res, dtype=object).ravel()
    for o in arr:
        if not isinstance(o, (Sym, int, float, np.integer, np.floating)):
            raise TraceError(f"{what}: the function returned {type(o).__name__}, not numbers")
    return list(arr)


def dynamics_source(fn, state_dim, noise_dim):
    """Record ``fn(x, q, u)`` -> (source text, output dimension)."""
    tape = _Tape()
    u = np.array([Sym(tape, "u")], dtype=object)
    outs = _flat(_record(fn, (_symbols(tape, "x", state_dim), _symbols(tape, "q", noise_dim), u), "dynamics_function"), "dynamics_function")
    return _finish(tape, outs, "template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out)"), len(outs)


def emission_source(fn, state_dim, noise_dim):
    tape = _Tape()
    u = np.array([Sym(tape, "u")], dtype=object)
    outs = _flat(_record(fn, (_symbols(tape, "x", state_dim), _symbols(tape, "r", noise_dim), u), "emission_function"), "emission_function")
    return _finish(tape, outs, "template <class T> __device__ void emission(const T* x, const T* r, T u, const float* th, T* out)"), len(outs)


def log_prob_source(fn, state_dim, emission_dim):
    tape = _Tape()
    u = np.array([Sym(tape, "u")], dtype=object)
    y = np.array([Sym(tape, f"T(y[{i}])") for i in range(emission_dim)], dtype=object)
    outs = _flat(_record(fn, (_symbols(tape, "x", state_dim), y, u), "emission_distribution_log_prob"), "emission_distribution_log_prob")
    if len(outs) != 1:
        raise TraceError("emission_distribution_log_prob must return one number")
    return _finish(tape, outs, "template <class T> __device__ T log_prob(const T* x, const float* y, T u, const float* th)", ret_scalar=True)
