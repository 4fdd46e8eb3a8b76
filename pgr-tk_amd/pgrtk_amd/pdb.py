"""`.pdb` cache of pgr-pbundle-decomp (pgr-bin/src/bin/pgr-pbundle-decomp.rs:155-218 reader, :357-383 writer):

    b"PDB:0.5" | bincode::encode_to_vec((w: u32, k: u32, r: u32, min_span: u32, min_branch_size: usize, min_cov: usize,
                                          principal_bundles_with_id: Vec<(usize, usize, Vec<(u64, u64, u8)>)>,
                                          vertex_to_bundle_id_direction_pos: HashMap<(u64, u64), (usize, u8, usize)>),
                                         bincode::config::standard())

bincode 2 `standard()` = little endian + variable-length integers: u8 is one raw byte; every other unsigned integer
(u16..u64, usize) is one byte when < 251, else a marker byte 251 / 252 / 253 followed by the value as u16 / u32 / u64
little endian (the smallest that fits); sequences and maps are their length (as u64, so varint) followed by the items;
tuples are their fields back to back.  bincode is not in /root/reference (a crates.io dependency, `bincode = "2.0.0-rc"`
in pgr-bin/Cargo.toml), so this is written from its published format description: PARITY UNPINNED against the reference's
own files (none are in its tree).  The map is written sorted by key (the reference writes hash-map order; readers do not
care)."""
import struct

MAGIC = b"PDB:0.5"


def _enc_uint(v, out):
    if v < 251:
        out.append(v)
    elif v < (1 << 16):
        out.append(251)
        out += struct.pack("<H", v)
    elif v < (1 << 32):
        out.append(252)
        out += struct.pack("<I", v)
    else:
        out.append(253)
        out += struct.pack("<Q", v)


class _Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def u8(self):
        v = self.d[self.p]
        self.p += 1
        return v

    def uint(self):
        m = self.u8()
        if m < 251:
            return m
        n, fmt = {251: (2, "<H"), 252: (4, "<I"), 253: (8, "<Q")}[m]
        v = struct.unpack_from(fmt, self.d, self.p)[0]
        self.p += n
        return v


def encode(w, k, r, min_span, min_branch_size, min_cov, bundles, vertex_map):
    """bundles: [(bundle_id, mean_order, [(h0, h1, direction)])]; vertex_map: {(h0, h1): (bundle_id, direction, position)}"""
    out = bytearray(MAGIC)
    for v in (w, k, r, min_span, min_branch_size, min_cov):
        _enc_uint(int(v), out)
    _enc_uint(len(bundles), out)
    for bid, order, verts in bundles:
        _enc_uint(int(bid), out)
        _enc_uint(int(order), out)
        _enc_uint(len(verts), out)
        for h0, h1, d in verts:
            _enc_uint(int(h0), out)
            _enc_uint(int(h1), out)
            out.append(int(d) & 0xFF)
    _enc_uint(len(vertex_map), out)
    for (h0, h1) in sorted(vertex_map):
        bid, d, pos = vertex_map[(h0, h1)]
        _enc_uint(int(h0), out)
        _enc_uint(int(h1), out)
        _enc_uint(int(bid), out)
        out.append(int(d) & 0xFF)
        _enc_uint(int(pos), out)
    return bytes(out)


def decode(data):
    """-> (w, k, r, min_span, min_branch_size, min_cov, bundles, vertex_map)"""
    if data[:7] != MAGIC:
        raise ValueError("not a PDB:0.5 file")
    rd = _Reader(data)
    rd.p = 7
    head = [rd.uint() for _ in range(6)]
    bundles = []
    for _ in range(rd.uint()):
        bid, order, n = rd.uint(), rd.uint(), rd.uint()
        bundles.append((bid, order, [(rd.uint(), rd.uint(), rd.u8()) for _ in range(n)]))
    vmap = {}
    for _ in range(rd.uint()):
        h0, h1, bid = rd.uint(), rd.uint(), rd.uint()
        d, pos = rd.u8(), rd.uint()
        vmap[(h0, h1)] = (bid, d, pos)
    if rd.p != len(data):
        raise ValueError("trailing bytes in the .pdb file")
    return tuple(head) + (bundles, vmap)


def write_pdb(path, *fields):
    with open(path, "wb") as f:
        f.write(encode(*fields))


def read_pdb(path):
    with open(path, "rb") as f:
        return decode(f.read())
