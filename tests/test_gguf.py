"""The GGUF reader + llama loader in front of the hot path (crabml_amd/csrc/host/gguf.hpp).

CPU tests: the container parser against files written by synth.write_gguf (v1 / v2 / v3 length widths, alignment key
of any integer type, data slices that run to the next tensor's offset, the "whole extra block" start of the data
section) and -- when /root/reference is present -- against the reference's own fixture with the reference's own
known answers (gguf.rs:838-965).  GPU tests: a model loaded from a GGUF file decodes bit-identically to the same
model uploaded tensor by tensor."""
import os
import struct

import numpy as np
import pytest

import crabml_amd as ca
from crabml_amd import synth

FIXTURE = "/root/reference/testdata/tinyllamas-stories-260k-f32.gguf"


def small_model(fmt=synth.Q4_0, **kw):
    return synth.build_model(synth.SHAPES["tiny-gqa"], fmt, seed=91, **kw)


@pytest.mark.parametrize("version", [1, 2, 3])
def test_parser_reads_what_the_writer_wrote(tmp_path, version):
    model = small_model(output_type=synth.Q6_K)
    path = str(tmp_path / "m.gguf")
    synth.write_gguf(model, path, version=version,
                     extra_kv=[("x.u8", "u8", 200), ("x.i8", "i8", -3), ("x.u16", "u16", 60000), ("x.i16", "i16", -30000),
                               ("x.i32", "i32", -7), ("x.u64", "u64", 2 ** 40), ("x.i64", "i64", -2 ** 40), ("x.f64", "f64", 0.25),
                               ("x.bool", "bool", 1), ("x.arr", "arr", ("i32", [1, -2, 3])),
                               ("x.nested", "arr", ("arr", [("u8", [1, 2]), ("u8", [])])),
                               ("x.strs", "arr", ("str", ["a", "", "héllo"]))])
    gf = ca.GGUFFile(path)
    assert gf.version == version and gf.architecture == "llama" and gf.alignment == 32
    md = gf.metadata()
    s = model.shape
    assert md["general.name"] == s.name
    assert md["llama.embedding_length"] == s.dim and md["llama.block_count"] == s.n_layers
    assert md["llama.attention.layer_norm_rms_epsilon"] == pytest.approx(1e-5)
    assert len(md["tokenizer.ggml.tokens"]) == s.vocab and md["tokenizer.ggml.tokens"][5] == "<5>"
    assert (md["x.u8"], md["x.i8"], md["x.u16"], md["x.i16"], md["x.i32"]) == (200, -3, 60000, -30000, -7)
    assert (md["x.u64"], md["x.i64"], md["x.f64"], md["x.bool"]) == (2 ** 40, -2 ** 40, 0.25, True)
    assert md["x.arr"] == [1, -2, 3] and md["x.nested"] == [[1, 2], []] and md["x.strs"] == ["a", "", "héllo"]
    infos = gf.tensor_infos()
    assert [i[0] for i in infos] == list(model.tensors.keys())
    for (name, dims, typ, off, nbytes), t in zip(infos, model.tensors.values()):
        assert dims == list(reversed(t.shape)) and typ == t.typ  # innermost first on disk (model.rs:473-475)
        assert off % 32 == 0
        data = gf.tensor_data(name)
        assert len(data) == nbytes and nbytes >= len(t.data) and nbytes - len(t.data) < 32  # padding rides along
        assert data[:len(t.data)] == t.data.tobytes()
    conf = gf.load_config()
    assert (conf.embedding_dim, conf.hidden_dim, conf.n_layers, conf.n_heads, conf.n_kv_heads, conf.vocab_size, conf.seq_len) == \
        (s.dim, s.hidden, s.n_layers, s.n_heads, s.n_kv_heads, s.vocab, s.seq_len)
    assert conf.rope_dim is None


@pytest.mark.parametrize("key_type,al", [("u32", 64), ("u64", 128), ("i32", 16), ("u8", 8), ("i64", 256)])
def test_alignment_key_of_any_integer_type(tmp_path, key_type, al):
    """gguf.rs:575-587; the data section starts at pos - pos % al + al (gguf.rs:722-724)."""
    model = small_model()
    path = str(tmp_path / "a.gguf")
    synth.write_gguf(model, path, alignment=al, write_alignment_key=key_type)
    gf = ca.GGUFFile(path)
    assert gf.alignment == al and gf.tensor_data_offset % al == 0
    name, t = next(iter(model.tensors.items()))
    assert gf.tensor_data(name)[:len(t.data)] == t.data.tobytes()
    last = list(model.tensors.keys())[-1]
    assert gf.tensor_data(last)[:64] == model.tensors[last].data.tobytes()[:64]


def test_malformed_files_are_format_errors(tmp_path):
    model = small_model()
    path = str(tmp_path / "ok.gguf")
    synth.write_gguf(model, path)
    raw = open(path, "rb").read()

    def bad(name, data):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        with pytest.raises(Exception):
            ca.GGUFFile(p)

    bad("magic.gguf", b"GGML" + raw[4:])                              # Invalid magic number (gguf.rs:525-527)
    bad("version.gguf", raw[:4] + struct.pack("<I", 9) + raw[8:])      # Unsupported version number (gguf.rs:529-536)
    bad("trunc.gguf", raw[:200])                                       # reads past the end (gguf.rs:260-272)
    noarch = synth.RawModel(model.shape, model.wtype, model.tensors)
    p = str(tmp_path / "noarch.gguf")
    synth.write_gguf(noarch, p)
    data = open(p, "rb").read().replace(b"general.architecture", b"general.architectur3")
    bad("noarch2.gguf", data)                                          # Missing string metadata general.architecture
    with pytest.raises(Exception):
        ca.GGUFFile(str(tmp_path / "does-not-exist.gguf"))


def test_missing_tensor_and_missing_keys(tmp_path):
    model = small_model()
    del model.tensors["blk.1.ffn_up.weight"]
    path = str(tmp_path / "t.gguf")
    synth.write_gguf(model, path)
    gf = ca.GGUFFile(path)
    with pytest.raises(Exception):
        gf.tensor_data("blk.1.ffn_up.weight")  # TensorNotFound (model.rs:486-491)
    assert gf.load_config().n_layers == model.shape.n_layers


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="reference fixture only exists in the build container")
def test_reference_fixture_known_answers():
    """The reference's own tests of its own fixture (gguf.rs:838-965), replayed against this parser."""
    gf = ca.GGUFFile(FIXTURE)
    infos = gf.tensor_infos()
    assert len(infos) == 48
    assert infos[0][0] == "token_embd.weight" and infos[0][4] == 131072 and infos[0][1] == [64, 512]
    assert all(i[2] == 0 for i in infos)  # F32
    exp = ["token_embd.weight - [64, 512]"]
    for l in range(5):
        b = f"blk.{l}."
        exp += [b + "attn_q.weight - [64, 64]", b + "attn_k.weight - [64, 32]", b + "attn_v.weight - [64, 32]",
                b + "attn_output.weight - [64, 64]", b + "ffn_gate.weight - [64, 172]", b + "ffn_down.weight - [172, 64]",
                b + "ffn_up.weight - [64, 172]", b + "attn_norm.weight - [64]", b + "ffn_norm.weight - [64]"]
    exp += ["output_norm.weight - [64]", "output.weight - [64, 512]"]
    assert [f"{i[0]} - {i[1]}" for i in infos] == exp
    assert gf.architecture == "llama" and gf.alignment == 32
    md = gf.metadata()
    assert sorted(md.keys()) == [
        "general.architecture", "general.name", "llama.attention.head_count", "llama.attention.head_count_kv",
        "llama.attention.layer_norm_rms_epsilon", "llama.block_count", "llama.context_length", "llama.embedding_length",
        "llama.feed_forward_length", "llama.rope.dimension_count", "llama.tensor_data_layout",
        "tokenizer.ggml.bos_token_id", "tokenizer.ggml.eos_token_id", "tokenizer.ggml.model",
        "tokenizer.ggml.padding_token_id", "tokenizer.ggml.scores", "tokenizer.ggml.token_type", "tokenizer.ggml.tokens"]
    assert md["general.name"] == "tinyllamas-stories-260k"
    assert (md["llama.attention.head_count"], md["llama.attention.head_count_kv"], md["llama.block_count"],
            md["llama.context_length"], md["llama.embedding_length"], md["llama.feed_forward_length"],
            md["llama.rope.dimension_count"]) == (8, 4, 5, 512, 64, 172, 8)
    assert md["llama.attention.layer_norm_rms_epsilon"] == pytest.approx(1e-5)
    assert md["llama.tensor_data_layout"] == "Meta AI original pth"
    assert gf.metadata_type("llama.block_count") == 4  # U32
    conf = gf.load_config()
    assert (conf.embedding_dim, conf.hidden_dim, conf.n_layers, conf.n_heads, conf.n_kv_heads, conf.vocab_size,
            conf.seq_len, conf.rope_dim) == (64, 172, 5, 8, 4, 512, 512, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,out", [("Q4_0", None), ("Q4_K", "Q6_K"), ("F32", None), ("Q8_0", "Q6_K"), ("Q5_K", None), ("Q3_K", "Q5_0"), ("Q2_K", "Q5_1")])
def test_gguf_loaded_model_decodes_like_the_uploaded_one(tmp_path, fmt, out):
    """llama.cpp-style files: layers of one type, token_embd / output of another; norm weights F32."""
    kw = dict(output_type=synth.TYPE_BY_NAME[out], embed_type=synth.TYPE_BY_NAME[out]) if out else {}
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=92, **kw)
    path = str(tmp_path / "m.gguf")
    synth.write_gguf(model, path, alignment=64, write_alignment_key="u32")
    dev = ca.HipTensorDevice(0, False, 0, True)
    conf_a, w_a = synth.to_hip(model, dev)
    conf_b, w_b = synth.load_gguf_hip(path, dev)
    assert conf_b.vocab_size == model.shape.vocab and conf_b.rms_norm_eps == pytest.approx(model.shape.rms_eps)
    a = ca.HipLlamaRunner(conf_a, w_a, dev, 32, True)
    b = ca.HipLlamaRunner(conf_b, w_b, dev, 32, True)
    for i, t in enumerate([1, 365, 400, 282, 7]):
        assert np.array_equal(a.forward(t, i).view(np.uint32), b.forward(t, i).view(np.uint32)), f"step {i}"
    # and through the trait-level runner (Llama2Runner<HipTensor>), which is what crabml-llama2 would drive
    c = ca.Llama2Runner(conf_b, w_b, dev, 32, True)
    d = ca.Llama2Runner(conf_a, w_a, dev, 32, True)
    for i, t in enumerate([1, 365, 400]):
        assert np.array_equal(np.asarray(c.forward([t], i)).view(np.uint32), np.asarray(d.forward([t], i)).view(np.uint32))


@pytest.mark.gpu
def test_gguf_f16_norm_weight_is_dequantized_on_load(tmp_path):
    """`.dequantize(GGMLType::F32)` of the norm weights (model.rs:267-281)."""
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q8_0, seed=93)
    ref = synth.RawModel(model.shape, model.wtype, dict(model.tensors))
    for name in list(model.tensors):
        if name.endswith("_norm.weight"):
            f = model.tensors[name].data.view(np.float32).astype(np.float16)
            model.tensors[name] = synth.RawTensor(f.view(np.uint8).copy(), model.tensors[name].shape, synth.F16)
            ref.tensors[name] = synth.RawTensor(f.astype(np.float32).view(np.uint8).copy(), model.tensors[name].shape, synth.F32)
    path = str(tmp_path / "n.gguf")
    synth.write_gguf(model, path)
    dev = ca.HipTensorDevice(0, False, 0, True)
    conf_a, w_a = synth.to_hip(ref, dev)
    conf_b, w_b = synth.load_gguf_hip(path, dev)
    a = ca.HipLlamaRunner(conf_a, w_a, dev, 16, True)
    b = ca.HipLlamaRunner(conf_b, w_b, dev, 16, True)
    for i, t in enumerate([1, 365, 400]):
        assert np.array_equal(a.forward(t, i).view(np.uint32), b.forward(t, i).view(np.uint32))


@pytest.mark.gpu
def test_gguf_q4_k_m_mix_round_trip(tmp_path):
    """A Q4_K_M-style file (per-tensor types differ inside a layer) through the GGUF loader into the fused step."""
    shape = synth.ModelShape("tiny-gqa-8l", 512, 1024, 8, 8, 2, 1024, 64)
    model = synth.build_model(shape, synth.Q4_K, seed=94, k_m_mix=True)
    path = str(tmp_path / "kmix.gguf")
    synth.write_gguf(model, path)
    gf = ca.GGUFFile(path)
    by_name = {i[0]: i[2] for i in gf.tensor_infos()}
    assert by_name["blk.0.ffn_down.weight"] == synth.Q6_K and by_name["blk.1.ffn_down.weight"] == synth.Q4_K
    dev = ca.HipTensorDevice(0, False, 0, True)
    conf_a, w_a = synth.to_hip(model, dev)
    conf_b, w_b = synth.load_gguf_hip(path, dev)
    a = ca.HipLlamaRunner(conf_a, w_a, dev, 32, True)
    b = ca.HipLlamaRunner(conf_b, w_b, dev, 32, True)
    for i, t in enumerate([1, 365, 400, 282]):
        assert np.array_equal(a.forward(t, i).view(np.uint32), b.forward(t, i).view(np.uint32)), f"step {i}"


@pytest.mark.gpu
def test_generate_tool_end_to_end(tmp_path):
    """tools/generate.py: GGUF file -> C++ loader -> batched prefill -> on-device greedy decode, as one command."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = small_model()
    path = str(tmp_path / "g.gguf")
    synth.write_gguf(model, path)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "generate.py"), path, "--steps", "6", "--prompt", "1,2,3,4,5"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("tokens:")][0]
    ids = [int(t) for t in line.split(":")[1].split(",")]
    assert len(ids) == 6 and all(0 <= t < model.shape.vocab for t in ids)
    # the same tokens through the python API
    dev = ca.HipTensorDevice(0)
    conf, w = synth.load_gguf_hip(path, dev)
    r = ca.HipLlamaRunner(conf, w, dev, 19, True)
    lg = r.prefill([1, 2, 3, 4, 5])
    first = int(len(lg) - 1 - lg[::-1].argmax())
    assert [first] + [int(t) for t in r.decode_greedy(first, 5)] == ids


@pytest.mark.gpu
def test_bench_runs_from_a_gguf_file(tmp_path):
    """bench.py --gguf FILE: the decode benchmark on a llama GGUF file (here a synthetic one) instead of in-memory weights."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=95, output_type=synth.Q6_K)
    path = str(tmp_path / "b.gguf")
    synth.write_gguf(model, path)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gguf", path, "--steps", "8", "--warmup", "2",
                          "--no-cpu-baseline", "--no-prefill"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["value"] > 0 and line["data"].startswith("file: ") and line["config"]["gemv_weight_bytes_per_token"] == model.gemv_weight_bytes_per_token()


def test_q4_k_m_recipe_layers_and_container_types(tmp_path):
    """CPU: the llama.cpp `use_more_bits` rule (first and last eighth of the layers, every third in between) and its
    round trip through the GGUF writer / reader (per-tensor ggml types)."""
    assert [i for i in range(32) if synth.use_more_bits(i, 32)] == [0, 1, 2, 3, 6, 9, 12, 15, 18, 21, 24, 27, 28, 29, 30, 31]
    assert [i for i in range(8) if synth.use_more_bits(i, 8)] == [0, 3, 6, 7]
    shape = synth.ModelShape("tiny-gqa-8l", 512, 1024, 8, 8, 2, 1024, 64)
    model = synth.build_model(shape, synth.Q4_K, seed=96, k_m_mix=True)
    path = str(tmp_path / "k.gguf")
    synth.write_gguf(model, path)
    types = {i[0]: i[2] for i in ca.GGUFFile(path).tensor_infos()}
    for l in range(8):
        wide = synth.Q6_K if synth.use_more_bits(l, 8) else synth.Q4_K
        assert types[f"blk.{l}.attn_v.weight"] == wide and types[f"blk.{l}.ffn_down.weight"] == wide
        assert types[f"blk.{l}.attn_q.weight"] == synth.Q4_K and types[f"blk.{l}.ffn_up.weight"] == synth.Q4_K
        assert types[f"blk.{l}.attn_norm.weight"] == synth.F32
    assert types["output.weight"] == synth.Q6_K and types["token_embd.weight"] == synth.Q4_K


# ---- robustness against crafted / unusual files -------------------------------------------------------------
def test_deeply_nested_arrays_raise_instead_of_overflowing_the_stack(tmp_path):
    """array-of-array-of-... 200k levels deep is 2.4 MB of file; the recursive reader used to die with SIGSEGV."""
    head = struct.pack("<II", 0x46554747, 3) + struct.pack("<QQ", 0, 1)
    key = b"x.deep"
    body = struct.pack("<Q", len(key)) + key + struct.pack("<I", 9)  # value type: array
    body += (struct.pack("<I", 9) + struct.pack("<Q", 1)) * 200000   # each level: elem type = array, length 1
    body += struct.pack("<I", 0) + struct.pack("<Q", 0)
    p = tmp_path / "deep.gguf"
    p.write_bytes(head + body)
    with pytest.raises(Exception) as ei:
        ca.GGUFFile(str(p))
    assert "nest" in str(ei.value)
    # four levels are fine (real files nest at most once)
    model = small_model()
    path = str(tmp_path / "ok.gguf")
    synth.write_gguf(model, path, extra_kv=[("x.n3", "arr", ("arr", [("arr", [("u8", [7])])]))])
    assert ca.GGUFFile(path).metadata()["x.n3"] == [[[7]]]


def test_array_length_is_bounded_by_the_bytes_that_are_left(tmp_path):
    head = struct.pack("<II", 0x46554747, 3) + struct.pack("<QQ", 0, 1)
    key = b"x.big"
    body = struct.pack("<Q", len(key)) + key + struct.pack("<I", 9) + struct.pack("<I", 10) + struct.pack("<Q", 1 << 40)  # 2^40 u64s
    p = tmp_path / "big.gguf"
    p.write_bytes(head + body + b"\0" * 64)
    with pytest.raises(Exception) as ei:
        ca.GGUFFile(str(p))
    assert "exceeds the file" in str(ei.value)


def _evil_file(tmp_path):
    """one F32 tensor whose dims (2^33, 2^33) multiply to 0 mod 2^64"""
    name = b"evil.weight"
    head = struct.pack("<II", 0x46554747, 3) + struct.pack("<QQ", 1, 1)
    k = b"general.architecture"
    head += struct.pack("<Q", len(k)) + k + struct.pack("<I", 8) + struct.pack("<Q", 5) + b"llama"
    head += struct.pack("<Q", len(name)) + name + struct.pack("<I", 2) + struct.pack("<QQ", 1 << 33, 1 << 33) + struct.pack("<IQ", 0, 0)
    pad = (32 - len(head) % 32) % 32
    p = tmp_path / "evil.gguf"
    p.write_bytes(head + b"\0" * pad + b"\0" * 64)
    return str(p)


def test_container_with_overflowing_dims_still_parses(tmp_path):
    gf = ca.GGUFFile(_evil_file(tmp_path))  # the container itself is well formed; the tensor is refused at load time
    assert gf.tensor_infos()[0][0] == "evil.weight"


@pytest.mark.gpu
def test_tensor_dims_whose_product_overflows_are_rejected_at_load(tmp_path):
    gf = ca.GGUFFile(_evil_file(tmp_path))
    dev = ca.HipTensorDevice(0)
    with pytest.raises(Exception) as ei:
        gf.load_tensor("evil.weight", dev)  # used to wrap to n = 0 and upload a 0-byte buffer carrying that shape
    assert "overflow" in str(ei.value)
    with pytest.raises(Exception) as ei:
        ca.HipTensor.from_cpu(np.zeros(16, np.uint8), [1 << 33, 1 << 33], ca.GGMLType.F32, dev)  # the C ABI checks, too
    assert "overflow" in str(ei.value)


@pytest.mark.parametrize("convention", ["spec", "reference"])
def test_header_that_ends_on_an_alignment_boundary(tmp_path, convention):
    """1 file in `alignment` ends its tensor infos exactly on a boundary.  The reference then skips a whole extra block
    (gguf.rs:722-724), the GGUF spec / llama.cpp writers do not pad at all: the reader resolves the case from the file
    (the data section must end where the last tensor ends) and reads the right bytes under BOTH conventions."""
    model = small_model()
    path = str(tmp_path / f"aligned-{convention}.gguf")
    synth.write_gguf(model, path, data_start=convention, pad_header_to_alignment=True)
    gf = ca.GGUFFile(path)
    assert gf.data_start_convention == (0 if convention == "spec" else 1)
    for name, t in model.tensors.items():
        assert gf.tensor_data(name)[:len(t.data)] == t.data.tobytes(), name
    # the independent python reader of the tests follows the spec
    if convention == "spec":
        from tests.helpers import read_gguf_py

        m2, _ = read_gguf_py(path)
        for name, t in model.tensors.items():
            assert np.array_equal(m2.tensors[name].data, t.data), name


def test_mlock_flag_and_madvise(tmp_path):
    model = small_model()
    path = str(tmp_path / "m.gguf")
    synth.write_gguf(model, path)
    try:
        gf = ca.GGUFFile(path, mlock=True)  # GGUFFileLoader::new(path, true) (gguf.rs:819-825)
    except Exception as e:  # RLIMIT_MEMLOCK may forbid it in a container: the error must say so
        assert "lock" in str(e)
    else:
        assert gf.load_config().n_layers == model.shape.n_layers


@pytest.mark.parametrize("convention", ["spec", "reference"])
def test_aligned_header_with_a_few_trailing_bytes_is_resolved_from_the_file(tmp_path, convention):
    """Round-3 advisor finding: a file whose tensor infos end on an alignment boundary and that carries a few trailing bytes was
    refused outright (the reference, gguf.rs:722-724, loads it).  A writer pads by less than one alignment block, so the right
    convention is the one under which the file ends < alignment bytes after the last tensor: both kinds load, with the right
    bytes."""
    model = small_model()
    path = str(tmp_path / f"aligned-trailing-{convention}.gguf")
    synth.write_gguf(model, path, data_start=convention, pad_header_to_alignment=True)
    with open(path, "ab") as f:
        f.write(b"\0" * 7)
    gf = ca.GGUFFile(path)
    assert gf.data_start_convention == (0 if convention == "spec" else 1)
    for name, t in model.tensors.items():
        assert gf.tensor_data(name)[:len(t.data)] == t.data.tobytes(), name


def test_aligned_header_with_a_block_of_trailing_bytes_is_refused_unless_told(tmp_path):
    """Round-2 review finding, narrowed: with a whole alignment block (or more) of trailing bytes the two conventions cannot be
    told apart from the file, and guessing wrong reads EVERY tensor one block off without any error downstream.  The reader
    refuses -- unless the caller names the convention (data_start), and then reads the right bytes."""
    model = small_model()
    for convention, code in (("spec", 0), ("reference", 1)):
        path = str(tmp_path / f"aligned-junk-{convention}.gguf")
        synth.write_gguf(model, path, data_start=convention, pad_header_to_alignment=True)
        with open(path, "ab") as f:
            f.write(b"\0" * 100)  # alignment is 32
        with pytest.raises(Exception) as ei:
            ca.GGUFFile(path)
        assert "ambiguous" in str(ei.value)
        gf = ca.GGUFFile(path, data_start=code)
        assert gf.data_start_convention == code
        for name, t in model.tensors.items():
            assert gf.tensor_data(name)[:len(t.data)] == t.data.tobytes(), name
