"""Reader of TensorFlow "TensorBundle" checkpoints (`<prefix>.index` + `<prefix>.data-00000-of-0000N`) without TensorFlow.

The reference's pretrained weights are such a checkpoint (`weights="detr"`: detr_tf/networks/weights.py:5-11 downloads
`detr.ckpt.index` / `detr.ckpt.data-00000-of-00001` and calls Keras `model.load_weights`, weights.py:33-34).  The format
(tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*):

  <prefix>.index   an SSTable (LevelDB table format): data blocks of prefix-compressed (key, value) entries with a restart
                   array, each followed by a 5-byte trailer (compression type 0 = none / 1 = snappy, masked crc32c), then a
                   metaindex block, an index block and a 48-byte footer (two BlockHandles, padding, magic 0xdb4775248b80fb57).
                   key ""  -> BundleHeaderProto  {num_shards = 1, endianness = 2, version = 3}
                   key k   -> BundleEntryProto   {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6, slices = 7}
  <prefix>.data-*  raw little-endian tensor bytes at (shard_id, offset, size); a DT_STRING tensor is stored as the varint
                   lengths of its elements, a 4-byte masked crc32c of those lengths, then the bytes.

A Keras object-based checkpoint (what `model.save_weights("x.ckpt")` writes) names its tensors by the path through the object
graph (`layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE`); the variables' own names live in the serialized
`TrackableObjectGraph` under the key `_CHECKPOINTABLE_OBJECT_GRAPH` (nodes[].attributes[]: name, full_name, checkpoint_key).
`load_tf_checkpoint` returns {variable full_name (":0" stripped) -> array}; name-based (V1) checkpoints, whose keys are the
variable names already, come back as they are.

Pinning: TensorFlow is not installable here, so the reader is tested against bundles written by the test suite's own
restatement of the writer side of the same format (tests/test_tf_checkpoint.py: uncompressed and snappy-compressed blocks,
multi-block indexes, prefix-compressed keys) -- parity with real TF-written files is UNPINNED until one can be read here.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
FOOTER_LEN = 48
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
VARIABLE_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"

# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DT_STRING, DT_BFLOAT16 = 7, 14


class CheckpointFormatError(ValueError):
    pass


# ---- primitives ----------------------------------------------------------------------------------------------------------------
def read_varint(buf, pos):
    """(value, new position) of a little-endian base-128 varint."""
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise CheckpointFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise CheckpointFormatError("varint longer than 64 bits")


_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of the table format and of the tensor bytes."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc):
    """leveldb / TF store crcs "masked": rotate right by 15 and add a constant (tensorflow/core/lib/hash/crc32c.h)."""
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def snappy_decompress(data):
    """Raw snappy block format: varint uncompressed length, then literal / copy elements (tag in the low two bits)."""
    n, pos = read_varint(data, 0)
    out = bytearray()
    while pos < len(data):
        tag = data[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += data[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                   # copy, 1-byte offset: length 4..11, offset 11 bits
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | data[pos]
            pos += 1
        elif kind == 2:                                 # copy, 2-byte offset
            ln = 1 + (tag >> 2)
            off = data[pos] | (data[pos + 1] << 8)
            pos += 2
        else:                                           # copy, 4-byte offset
            ln = 1 + (tag >> 2)
            off = int.from_bytes(data[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointFormatError("snappy: copy offset outside the output")
        for _ in range(ln):                             # (byte-wise: source and destination may overlap)
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointFormatError(f"snappy: {len(out)} bytes produced, header says {n}")
    return bytes(out)


# ---- minimal protobuf wire reader ------------------------------------------------------------------------------------------------
def parse_proto(buf):
    """[(field number, wire type, value)]: varint -> int, 64-bit / 32-bit -> raw bytes, length-delimited -> bytes."""
    out, pos = [], 0
    while pos < len(buf):
        key, pos = read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = read_varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = read_varint(buf, pos)
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise CheckpointFormatError(f"unsupported protobuf wire type {wt}")
        out.append((field, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def parse_bundle_entry(buf):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size, crc32c, sliced)."""
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for field, wt, v in parse_proto(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:                        # TensorShapeProto: repeated Dim dim = 2 {int64 size = 1}; unknown_rank = 3
            for f2, _, v2 in parse_proto(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in parse_proto(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e["shape"].append(size)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif field == 7:
            e["sliced"] = True
    return e


def parse_object_graph(buf):
    """TrackableObjectGraph -> [(attribute name, full_name, checkpoint_key)] of every saved attribute of every node."""
    out = []
    for f1, _, node in parse_proto(buf):
        if f1 != 1:
            continue
        for f2, _, attr in parse_proto(node):
            if f2 != 2:                         # children = 1, attributes = 2, slot_variables = 3
                continue
            name = full = key = ""
            for f3, wt, v in parse_proto(attr):
                if wt != 2:
                    continue
                if f3 == 1:
                    name = v.decode()
                elif f3 == 2:
                    full = v.decode()
                elif f3 == 3:
                    key = v.decode()
            out.append((name, full, key))
    return out


# ---- the table -------------------------------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    """Contents of the block at (offset, size): checks the trailer's masked crc32c and undoes the compression."""
    if offset + size + 5 > len(data):
        raise CheckpointFormatError("block handle points outside the file")
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        want = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
        if mask_crc(crc32c(data[offset:offset + size + 1])) != want:
            raise CheckpointFormatError(f"block at {offset}: checksum mismatch")
    if ctype == 0:
        return raw
    if ctype == 1:
        return snappy_decompress(raw)
    raise CheckpointFormatError(f"block at {offset}: unknown compression type {ctype}")


def _block_entries(block):
    """(key, value) pairs of a table block (keys are prefix-compressed against their predecessor; the restart array at the end
    only serves binary search)."""
    if len(block) < 4:
        raise CheckpointFormatError("block too short")
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    if end < 0:
        raise CheckpointFormatError("block: bad restart count")
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = read_varint(block, pos)
        non_shared, pos = read_varint(block, pos)
        vlen, pos = read_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise CheckpointFormatError("block: entry runs past the end")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path, verify=True):
    """All (key bytes, value bytes) of an SSTable file, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < FOOTER_LEN:
        raise CheckpointFormatError(f"{path}: too short for a table footer")
    footer = data[-FOOTER_LEN:]
    if struct.unpack("<Q", footer[-8:])[0] != TABLE_MAGIC:
        raise CheckpointFormatError(f"{path}: not a table file (bad magic number)")
    pos = 0
    _, pos = read_varint(footer, pos)           # metaindex handle (unused: no filter blocks in a bundle index)
    _, pos = read_varint(footer, pos)
    idx_off, pos = read_varint(footer, pos)
    idx_size, pos = read_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size, verify)):
        off, p = read_varint(handle, 0)
        size, p = read_varint(handle, p)
        out.extend(_block_entries(_read_block(data, off, size, verify)))
    return out


# ---- the bundle ------------------------------------------------------------------------------------------------------------------
def _shard_path(prefix, shard, num_shards):
    return f"{prefix}.data-{shard:05d}-of-{num_shards:05d}"


def read_bundle(prefix, verify_tensors=1 << 20):
    """{key: numpy array (or bytes / list of bytes for DT_STRING)} of every entry of the bundle `<prefix>.index`.
    Tensor crcs are checked for entries up to `verify_tensors` bytes (pure-Python CRC: the large tensors are skipped)."""
    index = prefix + ".index"
    if not os.path.exists(index):
        raise FileNotFoundError(index)
    entries = read_table(index)
    if not entries or entries[0][0] != b"":
        raise CheckpointFormatError(f"{index}: no bundle header entry")
    num_shards, endianness = 1, 0
    for field, _, v in parse_proto(entries[0][1]):
        if field == 1:
            num_shards = v
        elif field == 2:
            endianness = v
    if endianness != 0:
        raise CheckpointFormatError("big-endian bundles are not supported")
    shards = {}
    out = {}
    for key, val in entries[1:]:
        e = parse_bundle_entry(val)
        name = key.decode()
        if e["sliced"]:
            raise CheckpointFormatError(f"{name}: sliced (partitioned) variables are not supported")
        if e["shard_id"] not in shards:
            shards[e["shard_id"]] = np.memmap(_shard_path(prefix, e["shard_id"], num_shards), dtype=np.uint8, mode="r")
        raw = shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise CheckpointFormatError(f"{name}: data shard shorter than offset + size")
        if e["crc32c"] is not None and e["size"] <= verify_tensors and e["dtype"] != DT_STRING:
            if mask_crc(crc32c(bytes(raw))) != e["crc32c"]:
                raise CheckpointFormatError(f"{name}: tensor checksum mismatch")
        shape = tuple(e["shape"])
        if e["dtype"] == DT_STRING:
            n = int(np.prod(shape)) if shape else 1
            buf, pos, lens = bytes(raw), 0, []
            for _ in range(n):
                ln, pos = read_varint(buf, pos)
                lens.append(ln)
            pos += 4                                # masked crc32c of the lengths
            items = []
            for ln in lens:
                items.append(buf[pos:pos + ln])
                pos += ln
            out[name] = items[0] if not shape else items
        elif e["dtype"] == DT_BFLOAT16:
            u = np.frombuffer(bytes(raw), dtype="<u2").astype(np.uint32) << 16
            out[name] = u.view(np.float32).reshape(shape)
        elif e["dtype"] in DTYPES:
            out[name] = np.frombuffer(bytes(raw), dtype=np.dtype(DTYPES[e["dtype"]]).newbyteorder("<")).reshape(shape).copy()
        else:
            raise CheckpointFormatError(f"{name}: unsupported dtype {e['dtype']}")
    return out


def load_tf_checkpoint(prefix, duplicates=None):
    """{variable name: array}.  Object-based checkpoints are renamed through their object graph (attribute VARIABLE_VALUE ->
    the variable's full_name without the ':0' suffix); optimizer slots and bookkeeping entries (save counter) keep their
    checkpoint keys.  Name-based checkpoints are returned unchanged.
    Eager Keras variable names are not unique (several optimizers may each own `Adam/iter:0`): a repeated name RAISES unless the caller
    passes `duplicates` (a list): then the first entry of a name keeps it, later ones stay under their checkpoint key, and the list receives (name, key) of each
    -- the caller decides whether a duplicated name matters (networks.weights.load_tf_checkpoint_params raises when it is
    one of the parameters it was asked for)."""
    bundle = read_bundle(prefix)
    graph = bundle.pop(OBJECT_GRAPH_KEY, None)
    if graph is None:
        return {k: v for k, v in bundle.items() if isinstance(v, np.ndarray)}
    out = {}
    renamed = set()
    for name, full, key in parse_object_graph(graph):
        if name == "VARIABLE_VALUE" and key in bundle and full:
            var = full[:-2] if full.endswith(":0") else full
            if var in out:
                if duplicates is None:      # the lenient behaviour is opt-in: a caller that did not ask for the list must not map the wrong tensor silently
                    raise CheckpointFormatError(f"variable name {var!r} occurs more than once in the object graph (second checkpoint key {key!r}); "
                                                "pass duplicates=[] to keep the first and collect the others")
                duplicates.append((var, key))
                continue
            out[var] = bundle[key]
            renamed.add(key)
    for k, v in bundle.items():
        if k not in renamed and isinstance(v, np.ndarray):
            out[k] = v
    return out
