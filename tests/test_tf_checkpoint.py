"""CPU tests of the TensorFlow-checkpoint reader (detr_tf/networks/tf_checkpoint.py; SURVEY row N1, reference
detr_tf/networks/weights.py:5-37).  TensorFlow is not installable here, so the fixtures are bundles written by THIS file's
restatement of the writer side of the documented format (LevelDB table + BundleEntryProto + TrackableObjectGraph): what is
pinned is the reader against the format specification, not against bytes produced by TensorFlow itself."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))

from detr_tf.networks import tf_checkpoint as T  # noqa: E402


# ---- writer side (test infrastructure) -------------------------------------------------------------------------------------------
def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def pb_field(field, wt, payload):
    if wt == 0:
        return varint(field << 3) + varint(payload)
    if wt == 2:
        return varint((field << 3) | 2) + varint(len(payload)) + payload
    if wt == 5:
        return varint((field << 3) | 5) + payload
    raise ValueError(wt)


def snappy_literal_only(data):
    """A valid snappy stream made of literals only (what a compressor may always emit)."""
    out = bytearray(varint(len(data)))
    pos = 0
    while pos < len(data):
        chunk = data[pos:pos + 60]
        out.append((len(chunk) - 1) << 2)
        out += chunk
        pos += len(chunk)
    return bytes(out)


def table_block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", max(1, len(restarts)))
    return bytes(out)


def write_table(path, entries, block_entries=7, compress=False):
    """entries: sorted [(key bytes, value bytes)]; several data blocks + index block + empty metaindex block + footer."""
    f = bytearray()

    def emit(block):
        ctype = 0
        if compress:
            block, ctype = snappy_literal_only(block), 1
        off = len(f)
        f.extend(block)
        f.append(ctype)
        f.extend(struct.pack("<I", T.mask_crc(T.crc32c(block + bytes([ctype])))))
        return varint(off) + varint(len(block))

    index = []
    for i in range(0, len(entries), block_entries):
        chunk = entries[i:i + block_entries]
        index.append((chunk[-1][0] + b"\x00", emit(table_block(chunk))))     # any key >= the block's last key separates
    meta = emit(table_block([]))
    idx = emit(table_block(index, restart_interval=1))
    footer = meta + idx
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", T.TABLE_MAGIC)
    f.extend(footer)
    with open(path, "wb") as fh:
        fh.write(bytes(f))


DT = {np.dtype(np.float32): 1, np.dtype(np.int64): 9, np.dtype(np.int32): 3}


def write_bundle(prefix, tensors, object_graph=None, compress=False):
    """tensors: {checkpoint key: ndarray}; object_graph: [(attribute name, full_name, checkpoint_key)] or None."""
    data = bytearray()
    entries = [(b"", pb_field(1, 0, 1) + pb_field(3, 2, pb_field(1, 0, 1)))]          # header: num_shards 1, little endian, version
    items = {}
    for key, arr in tensors.items():
        raw = np.ascontiguousarray(arr).tobytes()
        shape = b"".join(pb_field(2, 2, pb_field(1, 0, d)) for d in arr.shape)
        e = pb_field(1, 0, DT[arr.dtype]) + pb_field(2, 2, shape) + pb_field(4, 0, len(data)) + pb_field(5, 0, len(raw)) + \
            pb_field(6, 5, struct.pack("<I", T.mask_crc(T.crc32c(raw))))
        data += raw
        items[key.encode()] = e
    if object_graph is not None:
        nodes = b""
        for name, full, key in object_graph:
            attr = pb_field(1, 2, name.encode()) + pb_field(2, 2, full.encode()) + pb_field(3, 2, key.encode())
            nodes += pb_field(1, 2, pb_field(1, 2, pb_field(1, 0, 0) + pb_field(2, 2, b"child")) + pb_field(2, 2, attr))
        lens = varint(len(nodes))
        raw = lens + struct.pack("<I", T.mask_crc(T.crc32c(lens))) + nodes                # DT_STRING scalar
        e = pb_field(1, 0, 7) + pb_field(2, 2, b"") + pb_field(4, 0, len(data)) + pb_field(5, 0, len(raw))
        data += raw
        items[T.OBJECT_GRAPH_KEY.encode()] = e
    entries += sorted(items.items())
    write_table(prefix + ".index", entries, compress=compress)
    with open(prefix + ".data-00000-of-00001", "wb") as fh:
        fh.write(bytes(data))


# ---- tests -------------------------------------------------------------------------------------------------------------------------
def test_primitives():
    for v in (0, 1, 127, 128, 300, 2 ** 32 + 5, 2 ** 63 - 1):
        assert T.read_varint(varint(v), 0) == (v, len(varint(v)))
    assert T.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    assert T.crc32c(b"") == 0
    # snappy: literal, 1-byte-offset copy (overlapping: run-length), 2-byte-offset copy, long literal
    blob = bytes(range(70))
    stream = varint(3 + 8 + 70 + 5) + bytes([2 << 2]) + b"abc" + bytes([((8 - 4) << 2) | 1, 3]) + \
        bytes([60 << 2, 69]) + blob + bytes([((5 - 1) << 2) | 2, 70, 0])
    assert T.snappy_decompress(stream) == b"abc" + b"abcabcab" + blob + blob[:5]
    assert T.snappy_decompress(snappy_literal_only(blob * 3)) == blob * 3
    with pytest.raises(T.CheckpointFormatError):
        T.snappy_decompress(varint(4) + bytes([((4 - 4) << 2) | 1, 9]))        # copy from before the start


@pytest.mark.parametrize("compress", [False, True])
def test_object_based_checkpoint_round_trip(tmp_path, compress):
    """A Keras-style object-based bundle (keys = object-graph paths, names in _CHECKPOINTABLE_OBJECT_GRAPH) over several table
    blocks, optionally snappy-compressed: every tensor comes back bit for bit under the variable's full name."""
    rng = np.random.default_rng(3)
    names = [f"detr/transformer/encoder/layer_{i}/linear1/{leaf}" for i in range(6) for leaf in ("kernel", "bias")] + \
            ["detr/backbone/conv1/kernel", "detr/query_embed/kernel", "save_counter"]
    shapes = {n: ((8, 4) if n.endswith("kernel") else (8,)) for n in names}
    shapes["detr/backbone/conv1/kernel"] = (7, 7, 3, 2)
    arrays = {n: rng.normal(size=shapes[n]).astype(np.float32) for n in names}
    arrays["save_counter"] = np.array(12, dtype=np.int64)
    keys = {n: f"layer_with_weights-{i}/w{i % 3}{T.VARIABLE_SUFFIX}" for i, n in enumerate(names)}
    graph = [("VARIABLE_VALUE", n + ":0", keys[n]) for n in names]
    prefix = str(tmp_path / "x.ckpt")
    write_bundle(prefix, {keys[n]: arrays[n] for n in names}, graph, compress=compress)
    got = T.load_tf_checkpoint(prefix)
    assert set(got) == set(names)
    for n in names:
        assert got[n].dtype == arrays[n].dtype and got[n].shape == arrays[n].shape and np.array_equal(got[n], arrays[n]), n
    raw = T.read_bundle(prefix)
    assert T.OBJECT_GRAPH_KEY in raw and isinstance(raw[T.OBJECT_GRAPH_KEY], bytes)


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "y.ckpt")
    write_bundle(prefix, {"a/kernel": np.arange(12, dtype=np.float32).reshape(3, 4)})
    assert np.array_equal(T.load_tf_checkpoint(prefix)["a/kernel"], np.arange(12, dtype=np.float32).reshape(3, 4))   # name-based form
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(5)
        f.write(b"\xff")
    with pytest.raises(T.CheckpointFormatError, match="tensor checksum"):
        T.load_tf_checkpoint(prefix)
    blob = bytearray(open(prefix + ".index", "rb").read())
    blob[3] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(blob))
    with pytest.raises(T.CheckpointFormatError, match="checksum mismatch"):
        T.read_table(prefix + ".index")
    open(prefix + ".index", "wb").write(b"not a table" * 8)
    with pytest.raises(T.CheckpointFormatError, match="magic"):
        T.read_table(prefix + ".index")


def test_whole_model_checkpoint_maps_onto_the_parameter_store(tmp_path):
    """Every trainable tensor and frozen-BN vector of a (small) DETR written as an object-based TF checkpoint under
    `detr/<reference layer name>`: networks.weights.load_tf_checkpoint_params maps all of them back (longest-suffix match:
    `.../layer1/0/conv1/kernel` vs the stem's `backbone/conv1/kernel`), shapes checked, nothing unused but the bookkeeping."""
    from detr_tf.networks.weights import load_tf_checkpoint_params
    from detr_tf.params import BN_LEAVES, bn_names, trainable_shapes
    blocks = (1, 1, 1, 1)
    wanted = dict(trainable_shapes(blocks, 1, 1, 10, 92, None))
    for p, c in bn_names(blocks).items():
        for leaf in BN_LEAVES[False]:
            wanted[f"{p}/{leaf}"] = (c,)
    rng = np.random.default_rng(5)
    arrays = {k: rng.normal(size=shp).astype(np.float32) for k, shp in wanted.items()}
    names = list(arrays)
    keys = {n: f"layer_with_weights-{i // 4}/v{i % 4}{T.VARIABLE_SUFFIX}" for i, n in enumerate(names)}
    tensors = {keys[n]: arrays[n] for n in names}
    tensors["save_counter" + T.VARIABLE_SUFFIX] = np.array(1, dtype=np.int64)
    graph = [("VARIABLE_VALUE", "detr/" + n + ":0", keys[n]) for n in names] + \
            [("VARIABLE_VALUE", "save_counter:0", "save_counter" + T.VARIABLE_SUFFIX)]
    prefix = str(tmp_path / "detr.ckpt")
    write_bundle(prefix, tensors, graph, compress=True)
    params, unused = load_tf_checkpoint_params(prefix, wanted)
    assert unused == ["save_counter"]
    assert set(params) == set(wanted)
    for k in wanted:
        assert np.array_equal(params[k], arrays[k]), k


@pytest.mark.gpu
def test_model_loads_a_tf_checkpoint_prefix(hip, tmp_path, monkeypatch):
    """`model.load_weights("<prefix>")` and `get_detr_model(weights="detr")` (files under weights/detr/, the reference's
    location, weights.py:24-33) read a TensorBundle written from another model's state: identical parameters, BN folded."""
    import torch
    from detr_tf.networks.detr import get_detr_model
    from detr_tf.training_config import TrainingConfig
    src = get_detr_model(TrainingConfig(), include_top=True, num_encoder_layers=1, num_decoder_layers=1, seed=11)
    state = src.engine.P.state_dict()
    names = list(state)
    keys = {n: f"layer_with_weights-{i}/v{T.VARIABLE_SUFFIX}" for i, n in enumerate(names)}
    graph = [("VARIABLE_VALUE", "detr/" + n + ":0", keys[n]) for n in names]
    (tmp_path / "weights" / "detr").mkdir(parents=True)
    prefix = str(tmp_path / "weights" / "detr" / "detr.ckpt")
    write_bundle(prefix, {keys[n]: state[n].astype(np.float32) for n in names}, graph)
    dst = get_detr_model(TrainingConfig(), include_top=True, num_encoder_layers=1, num_decoder_layers=1, seed=99)
    assert not torch.equal(dst.engine.P.flat, src.engine.P.flat)
    assert dst.load_weights(prefix) == []
    assert torch.equal(dst.engine.P.flat, src.engine.P.flat)
    monkeypatch.chdir(tmp_path)
    third = get_detr_model(TrainingConfig(), include_top=True, num_encoder_layers=1, num_decoder_layers=1, seed=5, weights="detr")
    assert torch.equal(third.engine.P.flat, src.engine.P.flat)
    x = np.random.default_rng(0).normal(size=(1, 64, 96, 3)).astype(np.float32)
    assert torch.equal(third(x)["pred_logits"], src(x)["pred_logits"])
    monkeypatch.chdir(tmp_path / "weights")
    with pytest.raises(FileNotFoundError, match="detr.ckpt.index"):
        get_detr_model(TrainingConfig(), include_top=True, num_encoder_layers=1, num_decoder_layers=1, weights="detr")


def test_duplicate_variable_names_only_matter_for_wanted_parameters(tmp_path):
    """Eager Keras names are not unique (two optimizers both own `Adam/iter:0`): such a checkpoint loads -- first entry under the
    name, later ones under their checkpoint keys -- and only a duplicated name that is one of the WANTED parameters is an error
    (ADVICE r4)."""
    from detr_tf.networks.weights import load_tf_checkpoint_params
    k = lambda i: f"obj-{i}/v{T.VARIABLE_SUFFIX}"
    arrays = {k(0): np.full((8, 4), 1.0, np.float32), k(1): np.array(3, np.int64), k(2): np.array(9, np.int64)}
    graph = [("VARIABLE_VALUE", "detr/class_embed/kernel:0", k(0)), ("VARIABLE_VALUE", "Adam/iter:0", k(1)),
             ("VARIABLE_VALUE", "Adam/iter:0", k(2))]
    prefix = str(tmp_path / "dup.ckpt")
    write_bundle(prefix, arrays, graph)
    dups = []
    got = T.load_tf_checkpoint(prefix, duplicates=dups)
    assert int(got["Adam/iter"]) == 3 and int(got[k(2)]) == 9 and dups == [("Adam/iter", k(2))]
    params, unused = load_tf_checkpoint_params(prefix, {"class_embed/kernel": (8, 4)})
    assert set(params) == {"class_embed/kernel"} and "Adam/iter" in unused
    arrays[k(3)] = np.full((8, 4), 2.0, np.float32)
    graph.append(("VARIABLE_VALUE", "detr/class_embed/kernel:0", k(3)))
    prefix2 = str(tmp_path / "dup2.ckpt")
    write_bundle(prefix2, arrays, graph)
    with pytest.raises(ValueError, match="two checkpoint entries"):
        load_tf_checkpoint_params(prefix2, {"class_embed/kernel": (8, 4)})
