"""oracle/pna_format.py -- TEST INFRASTRUCTURE ONLY (checker, never the product path).

CPU restatement of the PNA container as the reference writes/reads it.  Only chunk-level
Python loops (no per-byte loops); CRC-32 via zlib.  Every function cites the reference
file:line (relative to /root/reference) it follows.

Pinned by tests/test_oracle_container.py: byte-exact KATs from the reference's unit tests
(empty.pna, CRC known answers, header layouts, FlattenWriter / ChunkStreamWriter splitting)
and by re-serialising golden fixtures copied to tests/golden/ (deflate.pna round-trips
byte-exact, mirroring lib/tests/copy_entries.rs:15-21).
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Tuple

PNA_SIGNATURE = bytes([0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A])  # lib/src/format/signature.rs:6
MAX_CHUNK_DATA_LENGTH = 0xFFFFFFFF  # lib/src/chunk.rs:28

COMPRESSION_NO, COMPRESSION_DEFLATE, COMPRESSION_ZSTD, COMPRESSION_XZ = 0, 1, 2, 4  # entry/options.rs:241-247
KIND_FILE, KIND_DIR, KIND_SYMLINK, KIND_HARDLINK = 0, 1, 2, 3  # entry/options.rs:844-851
CIPHER_MODE_CBC, CIPHER_MODE_CTR = 0, 1  # entry/options.rs:600-604

CRITICAL_ENTRY_CHUNKS = {b"FHED", b"PHSF", b"FDAT", b"FEND", b"SHED", b"SDAT", b"SEND", b"AHED", b"AEND", b"ANXT"}


def chunk_crc(ty: bytes, data: bytes) -> int:
    """CRC-32 (IEEE) over type||data -- lib/src/format/chunk.rs:7-12."""
    return zlib.crc32(data, zlib.crc32(ty)) & 0xFFFFFFFF


def write_chunk(ty: bytes, data: bytes = b"") -> bytes:
    """len:u32be | type[4] | data | crc32:u32be -- lib/src/io.rs:183-197."""
    assert len(ty) == 4 and len(data) <= MAX_CHUNK_DATA_LENGTH
    return struct.pack(">I", len(data)) + ty + data + struct.pack(">I", chunk_crc(ty, data))


def archive_header_bytes(major: int = 0, minor: int = 0, archive_number: int = 0) -> bytes:
    """AHED body -- lib/src/archive/header.rs:27-39."""
    return bytes([major, minor, 0, 0]) + struct.pack(">I", archive_number)


def write_archive_header(archive_number: int = 0) -> bytes:
    """signature + AHED -- lib/src/archive.rs:21-25, lib/src/archive/write.rs:92-101."""
    return PNA_SIGNATURE + write_chunk(b"AHED", archive_header_bytes(0, 0, archive_number))


def finalize_archive() -> bytes:
    """AEND -- lib/src/archive/write.rs:438-440, lib/src/io.rs:45-51."""
    return write_chunk(b"AEND")


def entry_header_bytes(kind: int, compression: int, encryption: int, cipher_mode: int, name: str) -> bytes:
    """FHED body -- lib/src/entry/header.rs:123-134."""
    return bytes([0, 0, kind, compression, encryption, cipher_mode]) + name.encode("utf-8")


def file_entry_header(compression: int, name: str) -> bytes:
    """EntryHeader::for_file with no encryption: cipher_mode defaults to CTR(1)
    -- lib/src/entry/header.rs:55-62, lib/src/entry/options.rs:156-159."""
    return entry_header_bytes(KIND_FILE, compression, 0, CIPHER_MODE_CTR, name)


def dir_entry_header(name: str) -> bytes:
    """EntryHeader::for_dir: compression 0, encryption 0, cipher_mode CBC(0)
    -- lib/src/entry/header.rs:44-52,65-67."""
    return entry_header_bytes(KIND_DIR, 0, 0, CIPHER_MODE_CBC, name)


def solid_header_bytes(compression: int, encryption: int = 0, cipher_mode: int = CIPHER_MODE_CTR) -> bytes:
    """SHED body -- lib/src/entry/header.rs:274-282."""
    return bytes([0, 0, compression, encryption, cipher_mode])


def fsiz_bytes(raw_size: int) -> bytes:
    """u128 big-endian with leading zero bytes stripped (0 -> empty) -- lib/src/entry.rs:900-903."""
    return raw_size.to_bytes(16, "big").lstrip(b"\x00")


def sanitize_name(name: str) -> str:
    """EntryName::sanitize -- lib/src/entry/name.rs:148-156: normalize_utf8path first (lib/src/util/utf8path.rs:6-33: '.' dropped, '..'
    pops the preceding normal component, a '..' with nothing to pop is kept for now), then only Normal components survive, joined by '/'.
    '/' is the only separator off Windows (camino follows std::path)."""
    rooted = name.startswith("/")
    buf: List[str] = []
    for comp in name.split("/"):
        if comp in ("", "."):
            continue
        if comp == "..":
            if buf and buf[-1] != "..":
                buf.pop()
            elif not rooted:
                buf.append("..")          # Some(ParentDir) | None => push; with a root in front it is dropped
            continue
        buf.append(comp)
    return "/".join(c for c in buf if c != "..")


def flatten_writer(writes: Iterable[bytes], max_chunk_size: int = MAX_CHUNK_DATA_LENGTH) -> List[bytes]:
    """FlattenWriter: top-up last piece, then rest.chunks(max) -- lib/src/util/io.rs:60-77."""
    max_chunk_size = max(1, min(max_chunk_size, MAX_CHUNK_DATA_LENGTH))
    out: List[bytearray] = []
    for buf in writes:
        if not buf:
            continue
        rest = memoryview(buf)
        if out:
            free = max(0, max_chunk_size - len(out[-1]))
            head = rest[: min(len(rest), free)]
            out[-1] += head
            rest = rest[len(head):]
        for i in range(0, len(rest), max_chunk_size):
            out.append(bytearray(rest[i:i + max_chunk_size]))
    return [bytes(b) for b in out]


def chunk_stream_writer(ty: bytes, writes: Iterable[bytes], max_chunk_size: Optional[int] = None) -> bytes:
    """ChunkStreamWriter: each inner write -> chunk(s) of at most max -- lib/src/chunk/write.rs:32-47
    (write() emits one chunk of min(len,max) and returns that length; write_all loops)."""
    mx = max_chunk_size if max_chunk_size else 0xFFFFFFFF
    out = bytearray()
    for buf in writes:
        pos = 0
        while pos < len(buf):
            piece = buf[pos:pos + mx]
            out += write_chunk(ty, piece)
            pos += len(piece)
    return bytes(out)


def write_normal_entry(header: bytes, data_pieces: List[bytes], raw_file_size: Optional[int],
                       extra: Iterable[Tuple[bytes, bytes]] = (), facets: Iterable[Tuple[bytes, bytes]] = ()) -> bytes:
    """NormalEntry::write_chunks_to: FHED, extra*, fSIZ?, facets*, FDAT*, FEND -- lib/src/entry.rs:888-913."""
    out = bytearray(write_chunk(b"FHED", header))
    for ty, d in extra:
        out += write_chunk(ty, d)
    if raw_file_size is not None:
        out += write_chunk(b"fSIZ", fsiz_bytes(raw_file_size))
    for ty, d in facets:
        out += write_chunk(ty, d)
    for d in data_pieces:
        out += write_chunk(b"FDAT", d)
    out += write_chunk(b"FEND")
    return bytes(out)


def write_stream_entry(header: bytes, bursts: Iterable[bytes], extra: Iterable[Tuple[bytes, bytes]] = (),
                       facets: Iterable[Tuple[bytes, bytes]] = (), max_chunk_size: Optional[int] = None) -> bytes:
    """write_stream_entry -- lib/src/archive/write.rs:730-777: FHED, extra*, metadata facets*, then every burst the encoder hands to the
    ChunkStreamWriter as FDAT chunk(s) of at most max_chunk_size bytes (lib/src/chunk/write.rs:32-47), FEND.  No fSIZ: the size is not
    known when the header goes out (pinned by archive_write_file_accepts_attributes_without_generating_file_size, write.rs:830-881)."""
    out = bytearray(write_chunk(b"FHED", header))
    for ty, d in extra:
        out += write_chunk(ty, d)
    for ty, d in facets:
        out += write_chunk(ty, d)
    out += chunk_stream_writer(b"FDAT", bursts, max_chunk_size)
    out += write_chunk(b"FEND")
    return bytes(out)


def seek_to_end(buf: bytes) -> Tuple[int, bool]:
    """Archive::seek_to_end -- lib/src/archive/read.rs:412-424: skip chunks (no CRC check) up to AEND; returns (offset of the AEND chunk,
    whether an ANXT chunk was passed)."""
    if buf[:8] != PNA_SIGNATURE or buf[12:16] != b"AHED":
        raise ValueError("not a PNA archive")
    pos, nxt = 8, False
    while True:
        if len(buf) - pos < 12:
            raise ValueError("unexpected end of archive")
        (length,) = struct.unpack_from(">I", buf, pos)
        if len(buf) - pos - 12 < length:
            raise ValueError("unexpected end of archive")
        ty = bytes(buf[pos + 4:pos + 8])
        if ty == b"AEND":
            return pos, nxt
        if ty == b"ANXT":
            nxt = True
        pos += 12 + length


def write_encrypted_file_entry(compression: int, encryption: int, cipher_mode: int, name: str, phsf: str, iv: bytes,
                               ciphertext: bytes, raw_file_size: Optional[int], max_chunk_size: int = MAX_CHUNK_DATA_LENGTH) -> bytes:
    """A file entry written with a cipher: FHED (encryption, cipher_mode set), fSIZ, PHSF, FDAT(iv), FDAT(ciphertext)*, FEND.
    The IV is the data-stream prefix and becomes its own data piece: EntryBuilderCore::build -> prepend_data_prefix splices
    prefix.chunks(max) in front of the FlattenWriter pieces (lib/src/entry/builder.rs:62-69,171-188; prefix_bytes,
    lib/src/entry/write.rs:46-51); PHSF sits between the metadata and the data chunks (lib/src/entry.rs:905-910)."""
    header = entry_header_bytes(KIND_FILE, compression, encryption, cipher_mode, name)
    out = bytearray(write_chunk(b"FHED", header))
    if raw_file_size is not None:
        out += write_chunk(b"fSIZ", fsiz_bytes(raw_file_size))
    out += write_chunk(b"PHSF", phsf.encode("utf-8"))
    pieces = [iv[i:i + max_chunk_size] for i in range(0, len(iv), max_chunk_size)] + flatten_writer([ciphertext], max_chunk_size)
    for d in pieces:
        out += write_chunk(b"FDAT", d)
    out += write_chunk(b"FEND")
    return bytes(out)


def write_solid_entry(compression: int, sdat_pieces: List[bytes]) -> bytes:
    """SHED, SDAT*, SEND -- lib/src/archive/write.rs:443-470,716-727; lib/src/entry.rs:465-484."""
    out = bytearray(write_chunk(b"SHED", solid_header_bytes(compression)))
    for d in sdat_pieces:
        out += write_chunk(b"SDAT", d)
    out += write_chunk(b"SEND")
    return bytes(out)


# ----------------------------------------------------------------------------- reader

@dataclass
class ParsedEntry:
    kind: int
    compression: int
    encryption: int
    cipher_mode: int
    name: str
    raw_file_size: Optional[int]
    data: bytes                      # concatenated FDAT bodies (still compressed)
    chunks: List[Tuple[bytes, bytes]] = field(default_factory=list)


@dataclass
class ParsedSolid:
    compression: int
    encryption: int
    cipher_mode: int
    data: bytes                      # concatenated SDAT bodies
    chunks: List[Tuple[bytes, bytes]] = field(default_factory=list)


def read_chunks(buf: bytes, pos: int = 0):
    """read_chunk with mandatory CRC check -- lib/src/io.rs:117-149."""
    n = len(buf)
    while pos < n:
        if n - pos < 12:
            raise ValueError("truncated chunk header")
        (length,) = struct.unpack_from(">I", buf, pos)
        ty = bytes(buf[pos + 4:pos + 8])
        if n - pos - 12 < length:
            raise ValueError("truncated chunk body")
        data = bytes(buf[pos + 8:pos + 8 + length])
        (crc,) = struct.unpack_from(">I", buf, pos + 8 + length)
        if crc != chunk_crc(ty, data):
            raise ValueError(f"CRC mismatch in {ty!r}")
        pos += 12 + length
        yield ty, data, pos


def _is_critical(ty: bytes) -> bool:
    return (ty[0] & 0x20) == 0  # chunk/types.rs:52-57,291-299


def _entry_from_chunks(chunks: List[Tuple[bytes, bytes]]) -> ParsedEntry:
    """TryFrom<RawEntry> for NormalEntry -- lib/src/entry.rs:757-885."""
    ty0, hd = chunks[0]
    if ty0 != b"FHED":
        raise ValueError("entry must start with FHED")
    if len(hd) < 6 or hd[0] != 0 or hd[1] != 0:
        raise ValueError("unsupported entry version")
    size = None
    data = bytearray()
    for ty, d in chunks[1:]:
        if ty == b"FDAT":
            data += d
        elif ty == b"fSIZ":
            size = int.from_bytes(d, "big")
        elif ty in (b"FEND", b"PHSF"):
            pass
        elif _is_critical(ty):
            raise ValueError(f"unknown critical chunk {ty!r}")
    return ParsedEntry(hd[2], hd[3], hd[4], hd[5], hd[6:].decode("utf-8"), size, bytes(data), chunks)


def read_archive(buf: bytes):
    """Archive::read_header + next_raw_item loop -- lib/src/archive/read.rs:22-66.
    Returns (archive_number, [ParsedEntry | ParsedSolid])."""
    if buf[:8] != PNA_SIGNATURE:
        raise ValueError("bad signature")
    it = read_chunks(buf, 8)
    ty, d, _ = next(it)
    if ty != b"AHED" or len(d) != 8 or d[0] != 0:
        raise ValueError("first chunk must be AHED v0")
    archive_number = struct.unpack(">I", d[4:8])[0]
    items = []
    cur: List[Tuple[bytes, bytes]] = []
    ended = False
    for ty, d, _ in it:
        if ty == b"AEND":
            ended = True
            break
        cur.append((ty, d))
        if ty == b"FEND":
            items.append(_entry_from_chunks(cur)); cur = []
        elif ty == b"SEND":
            hd = cur[0][1]
            if cur[0][0] != b"SHED" or len(hd) != 5 or hd[0] != 0 or hd[1] != 0:
                raise ValueError("bad solid header")
            for t, _d in cur[1:]:
                if t not in (b"SDAT", b"SEND", b"PHSF") and _is_critical(t):
                    raise ValueError(f"unknown critical chunk {t!r}")
            sd = b"".join(x for t, x in cur if t == b"SDAT")
            items.append(ParsedSolid(hd[2], hd[3], hd[4], sd, cur)); cur = []
    if not ended or cur:
        raise ValueError("archive not terminated by AEND")
    return archive_number, items


def read_solid_inner(plain: bytes) -> List[ParsedEntry]:
    """read_next_normal_entry_from_stream over the decompressed SDAT stream -- lib/src/entry.rs:401-424."""
    entries = []
    cur: List[Tuple[bytes, bytes]] = []
    for ty, d, _ in read_chunks(plain, 0):
        cur.append((ty, d))
        if ty == b"FEND":
            entries.append(_entry_from_chunks(cur)); cur = []
    if cur:
        raise ValueError("dangling chunks in solid stream")
    return entries


# ----------------------------------------------------------------------------- multipart (lib/src/archive/split_parts.rs)

MIN_CHUNK_BYTES = 12                                   # lib/src/chunk.rs:24 (length + type + crc)
PART_HEADER_BYTES = 8 + MIN_CHUNK_BYTES + 8            # split_parts.rs:17: signature + AHED
SPLIT_ARCHIVE_OVERHEAD_BYTES = PART_HEADER_BYTES + 2 * MIN_CHUNK_BYTES   # :20: + ANXT + AEND
MIN_SPLIT_PART_BYTES = SPLIT_ARCHIVE_OVERHEAD_BYTES + MIN_CHUNK_BYTES    # :23


def split_parts(chunks: Iterable[Tuple[bytes, bytes]], max_part_bytes: int) -> List[bytes]:
    """SplitParts as a pure function: the (type, data) chunks of an archive body (everything between AHED and AEND) -> the bytes of
    every part.  put_chunk / put_stream / roll_over / write_part_framing / finalize_archive -- split_parts.rs:76-80,109-173,176-179,
    215-218: a chunk that fits goes out intact; a non-stream chunk that does not fit opens the next part; a stream chunk (FDAT / SDAT,
    chunk/types.rs:318-320) is cut at the budget boundary, its fragments framed and CRC'd anew."""
    if max_part_bytes < MIN_SPLIT_PART_BYTES:
        raise ValueError(f"max_part_bytes must be at least {MIN_SPLIT_PART_BYTES} bytes")
    budget = max_part_bytes - SPLIT_ARCHIVE_OVERHEAD_BYTES
    parts = [bytearray(write_archive_header(0))]
    remaining = budget

    def roll_over():
        nonlocal remaining
        parts[-1] += write_chunk(b"ANXT") + write_chunk(b"AEND")
        parts.append(bytearray(write_archive_header(len(parts))))      # `parts` doubles as the next part's archive number
        remaining = budget

    def put(ty, data):
        nonlocal remaining
        c = write_chunk(ty, data)
        parts[-1] += c
        remaining -= len(c)

    for ty, data in chunks:
        clen = MIN_CHUNK_BYTES + len(data)
        if clen <= remaining:
            put(ty, data); continue
        if ty not in (b"FDAT", b"SDAT"):
            if clen > budget:
                raise ValueError("chunk does not fit within the maximum part size")
            roll_over(); put(ty, data); continue
        if clen <= budget and remaining <= MIN_CHUNK_BYTES:
            roll_over(); put(ty, data); continue
        rest = data
        while True:                                                    # put_stream
            if MIN_CHUNK_BYTES + len(rest) <= remaining:
                put(ty, rest); break
            if remaining > MIN_CHUNK_BYTES:
                take = remaining - MIN_CHUNK_BYTES
                put(ty, rest[:take]); rest = rest[take:]
            elif budget <= MIN_CHUNK_BYTES:
                raise ValueError("chunk does not fit within the maximum part size")
            roll_over()
    parts[-1] += write_chunk(b"AEND")
    return [bytes(p) for p in parts]


def archive_body_chunks(buf: bytes) -> List[Tuple[bytes, bytes]]:
    """(type, data) of every chunk between AHED and AEND of one archive image (CRCs checked)."""
    if buf[:8] != PNA_SIGNATURE:
        raise ValueError("bad signature")
    out = []
    for i, (ty, d, _) in enumerate(read_chunks(buf, 8)):
        if i == 0:
            if ty != b"AHED":
                raise ValueError("first chunk must be AHED")
            continue
        if ty == b"AEND":
            return out
        out.append((ty, d))
    raise ValueError("archive not terminated by AEND")


def join_parts(parts: Iterable[bytes]) -> List[Tuple[bytes, bytes]]:
    """The reading side of a multipart archive (Archive::read_next_archive, lib/src/archive/read.rs): part k carries archive number k,
    every part but the last ends ANXT | AEND; the chunk streams concatenated are the archive body."""
    out = []
    parts = list(parts)
    for k, p in enumerate(parts):
        if p[:8] != PNA_SIGNATURE:
            raise ValueError("bad signature")
        it = read_chunks(p, 8)
        ty, d, _ = next(it)
        if ty != b"AHED" or struct.unpack(">I", d[4:8])[0] != k:
            raise ValueError("part out of order")
        body = [(t, x) for t, x, _ in it]
        if not body or body[-1][0] != b"AEND":
            raise ValueError("part not terminated by AEND")
        body.pop()
        has_next = bool(body) and body[-1][0] == b"ANXT"
        if has_next:
            body.pop()
        if has_next != (k + 1 < len(parts)):
            raise ValueError("ANXT does not match the number of parts")
        out += body
    return out
