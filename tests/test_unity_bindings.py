"""The C# P/Invoke shim a Unity maintainer would add (integration/unity/) against the C-ABI header it binds.

No C# compiler exists in the build image, so the check is textual: every function include/rt_native.h declares has
exactly one [DllImport] in RtNative.cs with the same name, the same number of arguments and argument / return types that
marshal to the C types; the mirrored structs have the header's fields; the wrapper classes expose the Unity method names
the reference calls (Main.cs:134-145,184-271,387-420, ShaderHelper.cs:7-73, ComputeHelper.cs:73-110,203-214) and call
only functions the shim declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITY = os.path.join(ROOT, "integration", "unity")

# C parameter type -> the C# declarations that marshal to it
MARSHAL = {
    "int": {"int"}, "float": {"float"}, "int64_t": {"long"}, "size_t": {"UIntPtr"}, "rt_format": {"int"},
    "rt_ctx": {"IntPtr"}, "rt_shader": {"IntPtr"}, "rt_buffer": {"IntPtr"}, "rt_texture": {"IntPtr"},
    "rt_ctx*": {"out IntPtr"}, "rt_shader*": {"out IntPtr"}, "rt_buffer*": {"out IntPtr"}, "rt_texture*": {"out IntPtr"},
    "const char*": {"string"}, "const void*": {"IntPtr"}, "void*": {"IntPtr"},
    "const int*": {"int[]"}, "const float[4]": {"float[]"}, "const float[16]": {"float[]"},
    "int*": {"out int"}, "double*": {"out double"},
    "rt_segment[4]": {"[Out] RtSegment[]"}, "rt_segment*": {"out RtSegment"},
    "rt_dispatch_stats*": {"out RtDispatchStats"}, "rt_kernel_profile*": {"out RtKernelProfile"},
}
RETURNS = {"int": "int", "void": "void", "const char*": "IntPtr", "void*": "IntPtr", "int64_t": "long"}


def header_functions():
    text = open(os.path.join(ROOT, "include", "rt_native.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"^((?:const )?\w+\*?) (rt_\w+)\(([^)]*)\);", text, flags=re.M):
        params = []
        for a in (x.strip() for x in args.split(",")):
            m = re.match(r"(.*?)(\w+)(\[\d+\])?$", a)          # type, name, optional array suffix
            params.append(m.group(1).strip().replace(" *", "*") + (m.group(3) or ""))
        out[name] = (ret, params)
    return out, text


def csharp_imports():
    text = open(os.path.join(UNITY, "RtNative.cs")).read()
    out = {}
    for ret, name, args in re.findall(r"\[DllImport\(Lib\)\] public static extern (\w+) (rt_\w+)\(([^;]*)\);", text):
        assert name not in out, f"{name} is imported twice"
        params = [re.sub(r"\s+\w+$", "", a.strip()) for a in args.split(",")] if args.strip() else []
        out[name] = (ret, params)
    return out, text


def test_every_header_function_has_a_matching_import():
    hdr, _ = header_functions()
    cs, _ = csharp_imports()
    assert len(hdr) >= 45, "the header parse lost declarations"
    assert sorted(hdr) == sorted(cs), (sorted(set(hdr) - set(cs)), sorted(set(cs) - set(hdr)))
    for name, (ret, params) in hdr.items():
        cret, cparams = cs[name]
        assert RETURNS[ret] == cret, f"{name}: returns {ret}, imported as {cret}"
        assert len(params) == len(cparams), f"{name}: {len(params)} arguments in the header, {len(cparams)} in the import"
        for i, (p, c) in enumerate(zip(params, cparams)):
            assert p in MARSHAL, f"{name}: no marshalling rule for '{p}'"
            assert c in MARSHAL[p], f"{name} argument {i}: '{p}' imported as '{c}'"


def test_structs_and_enums_mirror_the_header():
    _, h = header_functions()
    _, cs = csharp_imports()
    stats = re.findall(r"uint64_t (\w+);", re.search(r"typedef struct \{([^}]*)\} rt_dispatch_stats;", h).group(1))
    mine = re.search(r"public struct RtDispatchStats\s*\{\s*public ulong ([^;]*);", cs).group(1)
    assert [s.strip() for s in mine.split(",")] == stats
    assert re.search(r"public struct RtKernelProfile \{ public ulong launches; public double milliseconds; public RtDispatchStats stats; \}", cs)
    assert re.search(r"public struct RtSegment \{ public IntPtr ptr; public UIntPtr bytes; \}", cs)
    status = {k: int(v) for k, v in re.findall(r"RT_(?:ERR_)?(\w+) = (-?\d+)", re.search(r"typedef enum \{([^}]*)\} rt_status;", h).group(1))}
    mine = {k.upper(): int(v) for k, v in re.findall(r"(\w+) = (-?\d+)", re.search(r"public enum RtStatus \{([^}]*)\}", cs).group(1))}
    assert {k.replace("_", ""): v for k, v in status.items()} == mine
    fmt = {k.replace("_", ""): int(v) for k, v in re.findall(r"RT_FORMAT_(\w+) = (\d+)", h)}
    mine = {k.upper(): int(v) for k, v in re.findall(r"(\w+) = (\d+)", re.search(r"public enum RtFormat \{([^}]*)\}", cs).group(1))}
    assert fmt == mine


def test_wrappers_expose_the_unity_surface_the_reference_calls():
    declared, _ = csharp_imports()
    surface = {
        "NativeComputeShader.cs": ["FindKernel", "Dispatch", "SetInt", "SetFloat", "SetBool", "SetInts", "SetVector", "SetMatrix",
                                   "SetBuffer", "SetTexture", "EnableKeyword", "DisableKeyword"],
        "NativeComputeBuffer.cs": ["SetData", "GetData", "Release", "IsValid", "count", "stride"],
        "NativeRenderTexture.cs": ["CreateTexture", "Release", "GetPixels", "width", "height"],
        "NativeDevice.cs": ["Ctx", "Sync", "Shutdown"],
    }
    used = set()
    for fname, members in surface.items():
        text = open(os.path.join(UNITY, fname)).read()
        for m in members:
            assert re.search(rf"public [^;{{]*\b{m}\b", text), f"{fname} has no public {m}"
        for fn in re.findall(r"RtNative\.(rt_\w+)\(", text):
            assert fn in declared, f"{fname} calls {fn}, which RtNative.cs does not import"
            used.add(fn)
    # everything the reference's own calls need is reachable through the wrappers
    for fn in ("rt_create", "rt_destroy", "rt_shader_open", "rt_find_kernel", "rt_dispatch", "rt_set_int", "rt_set_float", "rt_set_bool",
               "rt_set_ints", "rt_set_vector", "rt_set_matrix", "rt_set_buffer", "rt_set_texture", "rt_enable_keyword", "rt_disable_keyword",
               "rt_buffer_create", "rt_buffer_set_data", "rt_buffer_get_data", "rt_buffer_release", "rt_texture_create",
               "rt_texture_set_data", "rt_texture_get_data", "rt_texture_release"):
        assert fn in used, f"no wrapper calls {fn}"
