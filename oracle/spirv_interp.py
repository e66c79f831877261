"""Minimal SPIR-V interpreter for the reference's four compute shaders.

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY.  The reference executes `shader/spv/*.comp.spv`
(they are `include_bytes!`'d: src/fft.rs:20-25, src/ocean.rs:26-28,195-197) on whatever
Vulkan/Metal/DX12 driver is present; none exists here, and neither do rustc/cargo/glslang.  This
file runs those *shipped binaries* instruction by instruction on numpy arrays (one array lane per
invocation, all invocations of a dispatch in lock-step, so barriers are trivially satisfied) with
the host-side wiring of src/render.rs:944-988 and the dispatch order of src/render.rs:1122-1310.
Its outputs on data/spectrum.bin + data/omega.bin are committed as tests/golden/spirv_*.npz
(generator: tests/golden/make_spirv_golden.py) and pin oracle/ocean_oracle.py: the literal
restatement must reproduce them to fp32 rounding.

What this is not: a Vulkan implementation.  It is the build's own reading of the SPIR-V
specification for the ~50 opcodes these shaders use; driver-defined precision (sin, cos, length,
division, FMA contraction) is fixed to "correctly rounded fp32, no contraction".  The .spv files
are read from /root/reference at run time and never copied into the repository.
"""
from __future__ import annotations

import struct

import numpy as np

F32, U32, I32 = np.float32, np.uint32, np.int32


class Ptr:
    """A pointer value: memory object + index path (ints or per-invocation index arrays)."""
    __slots__ = ("obj", "path")

    def __init__(self, obj, path=()):
        self.obj, self.path = obj, tuple(path)


class Mem:
    """A memory object.  scope: 'inv' (one copy per invocation), 'wg' (per workgroup), 'global'.
    data: ndarray, or list of ndarrays for a struct whose members differ in type."""

    def __init__(self, scope, data):
        self.scope, self.data = scope, data


class Module:
    def __init__(self, path):
        raw = open(path, "rb").read()
        w = struct.unpack("<%dI" % (len(raw) // 4), raw)
        if w[0] != 0x07230203:
            raise ValueError("not a SPIR-V module")
        self.types, self.consts, self.decor, self.member_decor = {}, {}, {}, {}
        self.globals, self.functions, self.names = {}, {}, {}
        self.local_size, self.entry = None, None
        i, cur = 5, None
        while i < len(w):
            wc, op = w[i] >> 16, w[i] & 0xFFFF
            a = w[i + 1:i + wc]
            i += wc
            if cur is not None:
                if op == 56:                           # OpFunctionEnd
                    cur = None
                elif op == 55:                         # OpFunctionParameter
                    cur["params"].append(a[1])
                elif op == 248:                        # OpLabel
                    cur["blocks"][a[0]] = []
                    cur["order"].append(a[0])
                    cur["cur"] = a[0]
                else:
                    cur["blocks"][cur["cur"]].append((op, a))
                continue
            if op == 5:
                self.names[a[0]] = self._str(a[1:])
            elif op == 15:
                self.entry = a[1]
            elif op == 16 and a[1] == 17:              # ExecutionMode LocalSize
                self.local_size = tuple(a[2:5])
            elif op == 71:
                self.decor.setdefault(a[0], {})[a[1]] = a[2:] or (1,)
            elif op == 72:
                self.member_decor.setdefault(a[0], {}).setdefault(a[1], {})[a[2]] = a[3:]
            elif op == 19:
                self.types[a[0]] = ("void",)
            elif op == 20:
                self.types[a[0]] = ("bool",)
            elif op == 21:
                self.types[a[0]] = ("int", a[1], a[2])
            elif op == 22:
                self.types[a[0]] = ("float", a[1])
            elif op == 23:
                self.types[a[0]] = ("vec", a[1], a[2])
            elif op == 25:
                self.types[a[0]] = ("image",)
            elif op == 28:
                self.types[a[0]] = ("array", a[1], a[2])       # length is a constant id
            elif op == 29:
                self.types[a[0]] = ("rtarray", a[1])
            elif op == 30:
                self.types[a[0]] = ("struct", tuple(a[1:]))
            elif op == 32:
                self.types[a[0]] = ("ptr", a[1], a[2])
            elif op == 33:
                self.types[a[0]] = ("fn",)
            elif op == 43:                             # OpConstant
                self.consts[a[1]] = self._scalar(a[0], a[2])
            elif op == 44:                             # OpConstantComposite
                self.consts[a[1]] = np.array([self.consts[c] for c in a[2:]])
            elif op == 59:                             # OpVariable (module scope)
                self.globals[a[1]] = (a[0], a[2])
            elif op == 54:                             # OpFunction
                cur = {"params": [], "blocks": {}, "order": [], "cur": None, "rtype": a[0]}
                self.functions[a[1]] = cur
            # 3 Source, 6 MemberName, 11 ExtInstImport, 14 MemoryModel, 17 Capability: ignored

    @staticmethod
    def _str(words):
        b = b"".join(struct.pack("<I", x) for x in words)
        return b.split(b"\0")[0].decode()

    def _scalar(self, tid, word):
        t = self.types[tid]
        if t[0] == "float":
            return np.frombuffer(struct.pack("<I", word), dtype=F32)[0]
        return (I32(np.array(word, U32).view(I32)) if t[2] else U32(word))

    def dtype_shape(self, tid, rt_len=None):
        t = self.types[tid]
        if t[0] == "float":
            return F32, ()
        if t[0] == "int":
            return (I32 if t[2] else U32), ()
        if t[0] == "bool":
            return np.bool_, ()
        if t[0] == "vec":
            d, _ = self.dtype_shape(t[1])
            return d, (t[2],)
        if t[0] == "array":
            d, s = self.dtype_shape(t[1])
            return d, (int(self.consts[t[2]]),) + s
        if t[0] == "rtarray":
            d, s = self.dtype_shape(t[1])
            return d, (rt_len,) + s
        raise NotImplementedError(t)


class Dispatch:
    """Executes one entry point for a whole dispatch (all invocations in lock-step)."""

    def __init__(self, mod: Module, groups, bindings):
        self.m = mod
        lx, ly, lz = mod.local_size
        gx, gy, gz = groups
        # invocation order: x fastest inside a workgroup, workgroups x fastest
        X, Y = gx * lx, gy * ly
        assert lz == 1 and gz == 1
        yy, xx = np.meshgrid(np.arange(Y, dtype=U32), np.arange(X, dtype=U32), indexing="ij")
        self.gid = np.stack([xx.ravel(), yy.ravel(), np.zeros(X * Y, U32)], axis=1)
        self.ninv = X * Y
        self.wg = ((yy.ravel() // ly) * gx + (xx.ravel() // lx)).astype(np.int64)
        self.nwg = gx * gy
        self.inv = np.arange(self.ninv)
        self.mem = {}
        for vid, (ptid, sc) in mod.globals.items():
            pointee = mod.types[ptid][2]
            dec = mod.decor.get(vid, {})
            if 11 in dec:                                   # BuiltIn
                assert dec[11][0] == 28                     # GlobalInvocationId
                self.mem[vid] = Mem("inv", self.gid.copy())
            elif sc == 4:                                   # Workgroup
                d, s = mod.dtype_shape(pointee)
                self.mem[vid] = Mem("wg", np.zeros((self.nwg,) + s, d))
            elif 33 in dec:                                 # Binding
                self.mem[vid] = bindings[dec[33][0]]
            else:
                raise NotImplementedError(("global", vid, sc))

    # -- memory -----------------------------------------------------------------------------------
    def _index(self, ptr: Ptr):
        obj, path = ptr.obj, list(ptr.path)
        data = obj.data
        if isinstance(data, list):                           # struct with heterogeneous members
            data = data[int(path.pop(0))]
            lead = ()
        elif obj.scope == "inv":
            lead = (self.inv,)
        elif obj.scope == "wg":
            lead = (self.wg,)
        else:
            lead = ()
        idx = lead + tuple(np.asarray(p).astype(np.int64) if isinstance(p, np.ndarray) else int(p) for p in path)
        return data, idx

    def load(self, ptr):
        if ptr.obj.scope == "image":                         # OpLoad of the image handle itself
            return ptr.obj.data
        data, idx = self._index(ptr)
        v = data[idx] if idx else data
        if not isinstance(v, np.ndarray) or v.shape[:1] != (self.ninv,):
            v = np.broadcast_to(v, (self.ninv,) + np.shape(v)).copy()
        return v

    def store(self, ptr, val):
        data, idx = self._index(ptr)
        if idx:
            data[idx] = val
        else:
            data[...] = val

    # -- execution ----------------------------------------------------------------------------------
    def run(self):
        self.call(self.m.entry, [])

    def call(self, fid, args):
        fn = self.m.functions[fid]
        val = dict(zip(fn["params"], args))
        m = self.m

        def V(i):
            if i in val:
                return val[i]
            if i in m.consts:
                return m.consts[i]
            if i in self.mem:
                return Ptr(self.mem[i])
            raise KeyError(i)

        def bc(x):                                            # broadcast a constant to [ninv, ...]
            x = np.asarray(x)
            if x.shape[:1] == (self.ninv,):
                return x
            return np.broadcast_to(x, (self.ninv,) + x.shape)

        label = fn["order"][0]
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            while True:
                for op, a in fn["blocks"][label]:
                    if op == 59:                              # OpVariable Function
                        d, s = m.dtype_shape(m.types[a[0]][2])
                        val[a[1]] = Ptr(Mem("inv", np.zeros((self.ninv,) + s, d)))
                    elif op == 61:
                        val[a[1]] = self.load(V(a[2]))
                    elif op == 62:
                        self.store(V(a[0]), bc(V(a[1])))
                    elif op == 65:                            # OpAccessChain
                        base = V(a[2])
                        val[a[1]] = Ptr(base.obj, base.path + tuple(V(i) for i in a[3:]))
                    elif op == 57:                            # OpFunctionCall
                        val[a[1]] = self.call(a[2], [V(i) for i in a[3:]])
                    elif op == 12:                            # OpExtInst GLSL.std.450
                        x = bc(V(a[4]))
                        if a[3] == 13:
                            r = np.sin(x.astype(np.float64)).astype(F32)
                        elif a[3] == 14:
                            r = np.cos(x.astype(np.float64)).astype(F32)
                        elif a[3] == 66:                      # Length: sqrt(x.x*x.x + x.y*x.y), fp32 steps
                            sq = (x * x).astype(F32)
                            acc = sq[:, 0]
                            for c in range(1, sq.shape[1]):
                                acc = (acc + sq[:, c]).astype(F32)
                            r = np.sqrt(acc).astype(F32)
                        else:
                            raise NotImplementedError(("GLSL.std.450", a[3]))
                        val[a[1]] = r
                    elif op == 79:                            # VectorShuffle
                        v = np.concatenate([bc(V(a[2])), bc(V(a[3]))], axis=1)
                        val[a[1]] = v[:, list(a[4:])]
                    elif op == 80:                            # CompositeConstruct
                        parts = [bc(V(i)) for i in a[2:]]
                        parts = [p[:, None] if p.ndim == 1 else p for p in parts]
                        val[a[1]] = np.concatenate(parts, axis=1)
                    elif op == 81:                            # CompositeExtract
                        v = bc(V(a[2]))
                        for ix in a[3:]:
                            v = v[:, ix]
                        val[a[1]] = v
                    elif op == 99:                            # ImageWrite
                        img, co, tx = V(a[0]), bc(V(a[1])), bc(V(a[2]))
                        img[co[:, 1].astype(np.int64), co[:, 0].astype(np.int64)] = tx
                    elif op == 112:                           # ConvertUToF (round to nearest even)
                        val[a[1]] = bc(V(a[2])).astype(U32).astype(F32)
                    elif op == 124:                           # Bitcast (same-width int <-> uint)
                        d, _ = m.dtype_shape(a[0])
                        val[a[1]] = np.ascontiguousarray(bc(V(a[2]))).view(d)
                    elif op == 127:
                        val[a[1]] = (-bc(V(a[2]))).astype(F32)
                    elif op in (128, 130, 132, 137, 196, 199):  # IAdd ISub IMul UMod Shl And (wrapping)
                        d, _ = m.dtype_shape(a[0])
                        x = np.ascontiguousarray(bc(V(a[2]))).view(U32)
                        y = np.ascontiguousarray(bc(V(a[3]))).view(U32)
                        r = {128: lambda: x + y, 130: lambda: x - y, 132: lambda: x * y, 137: lambda: x % y,
                             196: lambda: x << y, 199: lambda: x & y}[op]()
                        val[a[1]] = r.astype(U32).view(d)
                    elif op in (129, 131, 133, 136):          # FAdd FSub FMul FDiv (separately rounded)
                        x, y = bc(V(a[2])), bc(V(a[3]))
                        r = {129: np.add, 131: np.subtract, 133: np.multiply, 136: np.divide}[op](x, y)
                        val[a[1]] = r.astype(F32)
                    elif op == 142:                           # VectorTimesScalar
                        val[a[1]] = (bc(V(a[2])) * bc(V(a[3]))[:, None]).astype(F32)
                    elif op == 169:                           # Select
                        c, x, y = bc(V(a[2])), bc(V(a[3])), bc(V(a[4]))
                        val[a[1]] = np.where(c if x.ndim == 1 else c.reshape(-1, 1), x, y)
                    elif op == 170:
                        val[a[1]] = bc(V(a[2])).view(U32) == bc(V(a[3])).view(U32)
                    elif op == 176:
                        val[a[1]] = np.ascontiguousarray(bc(V(a[2]))).view(U32) < np.ascontiguousarray(bc(V(a[3]))).view(U32)
                    elif op == 186:
                        val[a[1]] = bc(V(a[2])) > bc(V(a[3]))
                    elif op in (224, 225, 246, 247):          # barriers / merge markers: lock-step => no-ops
                        pass
                    elif op == 249:                           # Branch
                        label = a[0]
                        break
                    elif op == 250:                           # BranchConditional: must be dispatch-uniform
                        c = bc(V(a[0]))
                        if c.all():
                            label = a[1]
                        elif not c.any():
                            label = a[2]
                        else:
                            raise NotImplementedError("divergent branch (not needed by these shaders)")
                        break
                    elif op == 253:
                        return None
                    elif op == 254:
                        return V(a[0])
                    else:
                        raise NotImplementedError(("opcode", op))
                else:
                    raise RuntimeError("block fell through without a terminator")


def run_reference_frame(spv_dir, h0, omega, time, resolution=512, domain_size=1000.0, return_stages=False):
    """The reference's per-frame sequence on its shipped SPIR-V: propagate [32,32,1]; fft_row and
    fft_col [1,512,1] on dx, dy, dz; correction [32,32,1]  (src/render.rs:1122-1310), with the
    descriptor wiring of src/render.rs:944-988.  N is the reference's hard-coded 512."""
    n = resolution
    assert n == 512 and h0.shape == (n, n)
    init = Mem("global", [np.ascontiguousarray(h0).view(F32).reshape(n * n, 2).copy()])
    om = Mem("global", [np.ascontiguousarray(omega, F32).reshape(n * n).copy()])
    dx, dy, dz = (Mem("global", [np.zeros((n * n, 2), F32)]) for _ in range(3))
    # PropagateLocals {time, resolution, domain_size}: src/render.rs:1107-1111
    locals_p = Mem("global", [np.array(time, F32), np.array(n, I32), np.array(domain_size, F32)])
    locals_c = Mem("global", [np.array(n, U32)])
    image = np.zeros((n, n, 4), F32)

    prop = Module(f"{spv_dir}/propagate.comp.spv")
    # bindings 1=initial_spec 2=omega 3=dy_spec(height) 4=dx_spec 5=dz_spec: src/render.rs:944-957
    Dispatch(prop, (n // 16, n // 16, 1), {0: locals_p, 1: init, 2: om, 3: dy, 4: dx, 5: dz}).run()
    stages = {"propagate": tuple(b.data[0].copy() for b in (dy, dx, dz))}
    for name in ("fft_row", "fft_col"):
        mod = Module(f"{spv_dir}/{name}.comp.spv")
        for buf in (dx, dy, dz):                           # desc_sets[0,1,2]: src/render.rs:971-988
            Dispatch(mod, (1, n, 1), {0: buf}).run()
        stages[name] = tuple(b.data[0].copy() for b in (dy, dx, dz))
    cor = Module(f"{spv_dir}/correction.comp.spv")
    # bindings 1=dy_spec 2=dx_spec 3=dz_spec 4=image: src/render.rs:958-970
    Dispatch(cor, (n // 16, n // 16, 1), {0: locals_c, 1: dy, 2: dx, 3: dz, 4: Mem("image", image)}).run()
    return (image, stages) if return_stages else image
