"""The Rust side of the boundary, as files (north_star: "a new `crabml-hip` crate ... Rust host code calling hand-written
HIP kernels through a thin C-ABI layer").  There is no rustc in this image, so the crate cannot be compiled here; what CAN be
checked on a CPU is that it stays in step with the library it binds:

  * crabml-hip/src/ffi.rs declares exactly the functions include/crabml_hip.h declares -- same names, same number of
    arguments, same integer widths / pointer constness per argument and for the return type;
  * the #[repr(C)] structs have the header's fields, in order, with matching types;
  * the status -> ErrorKind table of hip_device.rs matches the header's enum (= crabml-core/src/error.rs:5-33);
  * every method of the `Tensor` trait (crabml-core/src/tensor/api.rs:11-79) is implemented by HipTensor;
  * the two patches against the reference (quantized weights reach a device backend: model.rs:817-837; `-D hip` in the
    CLI: main.rs:66-79, 248-263) still apply to /root/reference (build container only)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "crabml_hip.h")
CRATE = os.path.join(ROOT, "crabml-hip")
REFERENCE = "/root/reference"


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


C_SCALARS = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "size_t": "usize", "float": "f32", "double": "f64",
             "uint64_t": "u64", "char": "c_char", "void": "c_void"}


def c_type_to_rust(t):
    """'const crabml_hip_buf_t* const*' -> '*const *const crabml_hip_buf_t' (canonical, whitespace-free tokens)"""
    t = t.strip()
    toks = re.findall(r"\*|const|[A-Za-z_][A-Za-z0-9_]*", t)
    # base type = first identifier that is not const
    base = next(x for x in toks if x not in ("const", "*"))
    base_const = "const" in toks[:toks.index(base) + 2] and (toks.index("const") < toks.index(base) or toks[toks.index(base) + 1:toks.index(base) + 2] == ["const"])
    rest = toks[toks.index(base) + 1:]
    if rest[:1] == ["const"]:
        rest = rest[1:]
    ptrs = []  # constness of what each successive '*' points to
    pointee_const = base_const
    for tok in rest:
        if tok == "*":
            ptrs.append(pointee_const)
            pointee_const = False
        elif tok == "const":
            pointee_const = True
    out = C_SCALARS.get(base, base)
    for is_const in ptrs:
        out = ("*const " if is_const else "*mut ") + out
    return out


def parse_header():
    src = strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(crabml_hip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        arg_types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)  # strip the parameter name
                arg_types.append(c_type_to_rust(mm.group(1)))
        protos[name] = (c_type_to_rust(ret), arg_types)
    structs = {}
    for m in re.finditer(r"typedef struct (crabml_hip_[a-z_]+)\s*\{(.*?)\}\s*(crabml_hip_[a-z_]+_t)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*(?:\s*,\s*[A-Za-z_][A-Za-z0-9_]*)*)$", decl)
            typ = c_type_to_rust(mm.group(1))
            for nm in mm.group(2).split(","):
                fields.append((nm.strip(), typ))
        structs[m.group(3)] = fields
    return protos, structs


def parse_ffi():
    src = re.sub(r"//.*", "", open(os.path.join(CRATE, "src", "ffi.rs")).read())
    block = re.search(r'extern "C" \{(.*)\}', src, flags=re.S).group(1)
    protos = {}
    for m in re.finditer(r"pub fn (crabml_hip_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2).strip(), (m.group(3) or "()").strip()
        arg_types = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in args.split(",") if a.strip()]
        protos[name] = (re.sub(r"\s+", " ", ret), arg_types)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (crabml_hip_[a-z_]+_t)\s*\{(.*?)\}", src, flags=re.S):
        fields = [(f.split(":", 1)[0].replace("pub", "").strip(), re.sub(r"\s+", " ", f.split(":", 1)[1].strip()))
                  for f in m.group(2).split(",") if ":" in f]
        structs[m.group(1)] = fields
    return protos, structs


def test_ffi_declares_exactly_the_functions_of_the_header():
    hp, _ = parse_header()
    fp, _ = parse_ffi()
    assert len(hp) >= 44  # the drop-in surface (the hooks of crabml_hip_debug.h are not bound by the crate)
    assert sorted(hp) == sorted(fp), (sorted(set(hp) - set(fp)), sorted(set(fp) - set(hp)))
    for name, (ret, args) in hp.items():
        fret, fargs = fp[name]
        assert len(args) == len(fargs), f"{name}: {len(args)} arguments in the header, {len(fargs)} in ffi.rs"
        assert ret == fret, f"{name}: returns {ret} in the header, {fret} in ffi.rs"
        for i, (a, b) in enumerate(zip(args, fargs)):
            assert a == b, f"{name}: argument {i} is {a} in the header, {b} in ffi.rs"


def test_repr_c_structs_match_the_header():
    _, hs = parse_header()
    _, fs = parse_ffi()
    for name in ("crabml_hip_device_options_t", "crabml_hip_llama_config_t", "crabml_hip_llama_weights_t"):
        assert name in hs and name in fs, name
        assert [f[0] for f in hs[name]] == [f[0] for f in fs[name]], name
        for (hn, ht), (fn, ft) in zip(hs[name], fs[name]):
            assert ht == ft, f"{name}.{hn}: {ht} in the header, {ft} in ffi.rs"


def test_status_codes_map_onto_error_kind_in_order():
    src = strip_c_comments(open(HEADER).read())
    enum = re.search(r"typedef enum crabml_hip_status \{(.*?)\}", src, flags=re.S).group(1)
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"CRABML_HIP_([A-Z_]+)\s*=\s*(\d+)", enum)}
    rs = open(os.path.join(CRATE, "src", "hip_device.rs")).read()
    table = dict((int(a), b) for a, b in re.findall(r"(\d+) => ErrorKind::(\w+)", rs))
    want = {"IO_ERROR": "IOError", "TENSOR_NOT_FOUND": "TensorNotFound", "MODEL_ERROR": "ModelError", "BAD_INPUT": "BadInput",
            "FORMAT_ERROR": "FormatError", "TENSOR_ERROR": "TensorError", "CHAT_TEMPLATE_NOT_FOUND": "ChatTemplateNotFound",
            "NOT_IMPLEMENTED": "NotImplemented"}
    for cname, kind in want.items():
        assert table[codes[cname]] == kind, cname
    assert codes["OK"] == 0 and codes["UNEXPECTED"] == 1 and "_ => ErrorKind::Unexpected" in rs
    if os.path.exists(os.path.join(REFERENCE, "crabml-core", "src", "error.rs")):
        kinds = re.findall(r"(?m)^    (\w+),$", re.search(r"pub enum ErrorKind \{(.*?)\n\}", open(
            os.path.join(REFERENCE, "crabml-core", "src", "error.rs")).read(), flags=re.S).group(1))
        # status = discriminant + 1
        for i, k in enumerate(kinds):
            assert (table.get(i + 1) or "Unexpected") == k, (i, k)


def test_ggml_type_ids_are_passed_as_the_enum_discriminants():
    src = strip_c_comments(open(HEADER).read())
    enum = re.search(r"typedef enum crabml_hip_ggml_type \{(.*?)\}", src, flags=re.S).group(1)
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r"CRABML_HIP_([A-Z0-9_]+)\s*=\s*(\d+)", enum)}
    assert ids == {"F32": 0, "F16": 1, "Q4_0": 2, "Q4_1": 3, "Q5_0": 6, "Q5_1": 7, "Q8_0": 8, "Q8_1": 9, "Q2_K": 10, "Q3_K": 11, "Q4_K": 12,
                   "Q5_K": 13, "Q6_K": 14, "Q8_K": 15}
    assert "dtype as u32" in open(os.path.join(CRATE, "src", "hip_tensor.rs")).read()
    if os.path.exists(os.path.join(REFERENCE, "crabml-core", "src", "gguf.rs")):
        g = open(os.path.join(REFERENCE, "crabml-core", "src", "gguf.rs")).read()
        enum = re.search(r"pub enum GGMLType \{(.*?)\n\}", g, flags=re.S).group(1)
        ref = {m.group(1): int(m.group(2)) for m in re.finditer(r"(\w+) = (\d+),", enum)}
        for c_name, r_name in (("F32", "F32"), ("F16", "F16"), ("Q4_0", "Q4_0"), ("Q4_1", "Q4_1"), ("Q8_0", "Q8_0"), ("Q8_1", "Q8_1"),
                               ("Q4_K", "Q4K"), ("Q5_K", "Q5K"), ("Q6_K", "Q6K"), ("Q8_K", "Q8K"), ("Q5_0", "Q5_0"), ("Q5_1", "Q5_1"),
                               ("Q2_K", "Q2K"), ("Q3_K", "Q3K")):
            assert ids[c_name] == ref[r_name]


def test_hip_tensor_implements_every_method_of_the_tensor_trait():
    want = ["from_cpu", "alloc", "resize", "dtype", "with_strider", "with_name", "reshape", "transpose", "contiguous", "shape",
            "strider", "concatenate", "copy_rows_from", "export", "dup", "rope_inplace", "rms_norm_inplace", "softmax_inplace",
            "silu_inplace", "gelu_inplace", "mul_inplace", "add_inplace", "scale_inplace", "matmul_vec", "batch_matmul"]
    api = os.path.join(REFERENCE, "crabml-core", "src", "tensor", "api.rs")
    if os.path.exists(api):
        trait = re.search(r"pub trait Tensor.*?\{(.*)\}", open(api).read(), flags=re.S).group(1)
        assert re.findall(r"fn (\w+)", trait) == want
    rs = open(os.path.join(CRATE, "src", "hip_tensor.rs")).read()
    body = rs[rs.index("impl Tensor for HipTensor"):rs.index("#[cfg(test)]")]
    got = re.findall(r"(?m)^    fn (\w+)", body)
    assert sorted(got) == sorted(want), (set(want) - set(got), set(got) - set(want))
    assert "type DeviceRef = HipTensorDeviceRef;" in body
    # no elided bodies: every method ends in real code
    assert "/* ..." not in rs and "todo!" not in rs and "unimplemented!" not in rs
    for f in ("Cargo.toml", "build.rs", "src/lib.rs", "src/ffi.rs", "src/hip_device.rs", "src/hip_tensor.rs", "src/hip_llama.rs"):
        assert os.path.getsize(os.path.join(CRATE, f)) > 200, f


@pytest.mark.skipif(not os.path.isdir(REFERENCE) or shutil.which("git") is None, reason="needs the reference checkout (build container)")
@pytest.mark.parametrize("patch", ["0001-quantized-device-weights.patch", "0002-cli-device-hip.patch"])
def test_patches_apply_to_the_reference(patch):
    r = subprocess.run(["git", "apply", "--check", "-p1", os.path.join(ROOT, "patches", patch)], cwd=REFERENCE, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.isdir(REFERENCE) or shutil.which("git") is None, reason="needs the reference checkout (build container)")
def test_patched_reference_has_the_hip_arm_and_passes_the_stored_type(tmp_path):
    work = tmp_path / "ref"
    shutil.copytree(REFERENCE, work, ignore=shutil.ignore_patterns("testdata", ".git", "target"))
    for root, dirs, files in os.walk(work):
        for d in dirs:
            os.chmod(os.path.join(root, d), 0o755)
        for f in files:
            os.chmod(os.path.join(root, f), 0o644)
    for patch in ("0001-quantized-device-weights.patch", "0002-cli-device-hip.patch"):
        r = subprocess.run(["git", "apply", "-p1", os.path.join(ROOT, "patches", patch)], cwd=work, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    model = open(work / "crabml-llama2" / "src" / "model.rs").read()
    assert "tensor.dtype(), device.clone()" in model and "unsupported tensor type on gpu" not in model
    main = open(work / "crabml-cli" / "src" / "main.rs").read()
    assert "DeviceType::Hip =>" in main and "GpuLlamaModel::<HipTensor>::from_cpu" in main
    assert '"crabml-hip",' in open(work / "Cargo.toml").read()
