"""Minimal TensorBoard scalar writer (no tensorboard / tensorflow needed).

Ray Tune of the reference's era logged every numeric result column as a scalar under ``ray/tune/<key>`` into an
``events.out.tfevents.*`` file of the trial directory (that is what ``tensorboard --logdir ~/ray_results`` of the
reference's README shows).  This writes the same thing: TFRecord framing (length, masked CRC32C of the length, payload,
masked CRC32C of the payload) around hand-encoded ``Event`` protobufs
(wall_time = 1: double, step = 2: int64, file_version = 3: string, summary = 5 { value = 1 { tag = 1, simple_value = 2 } }).
"""
import os
import socket
import struct
import time

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num: int, wire: int, payload: bytes) -> bytes:
    return _varint((num << 3) | wire) + payload


def _bytes_field(num: int, data: bytes) -> bytes:
    return _field(num, 2, _varint(len(data)) + data)


def encode_event(wall_time: float, step: int = 0, scalars=None, file_version: str = None) -> bytes:
    ev = _field(1, 1, struct.pack("<d", wall_time)) + _field(2, 0, _varint(step))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if scalars:
        summary = b""
        for tag, value in scalars:
            val = _bytes_field(1, tag.encode()) + _field(2, 5, struct.pack("<f", float(value)))
            summary += _bytes_field(1, val)
        ev += _bytes_field(5, summary)
    return ev


def frame(payload: bytes) -> bytes:
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload))


class ScalarWriter:
    def __init__(self, logdir: str):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}")
        self._f = open(self.path, "ab")
        self._f.write(frame(encode_event(time.time(), 0, file_version="brain.Event:2")))
        self._f.flush()

    def add_scalars(self, scalars: dict, step: int, prefix: str = "ray/tune/") -> None:
        items = [(prefix + k, v) for k, v in scalars.items()
                 if isinstance(v, (int, float)) and not isinstance(v, bool) and v == v]
        if items:
            self._f.write(frame(encode_event(time.time(), step, items)))
            self._f.flush()

    def close(self) -> None:
        self._f.close()


def read_events(path: str):
    """Parse a file written by ScalarWriter back into [(step, {tag: value})] (checks every CRC)."""
    out = []
    data = open(path, "rb").read()
    pos = 0

    def varint(buf, p):
        n, shift = 0, 0
        while True:
            b = buf[p]; p += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, p

    def fields(buf):
        p = 0
        while p < len(buf):
            key, p = varint(buf, p)
            num, wire = key >> 3, key & 7
            if wire == 0:
                v, p = varint(buf, p)
            elif wire == 1:
                v = buf[p:p + 8]; p += 8
            elif wire == 5:
                v = buf[p:p + 4]; p += 4
            else:
                n, p = varint(buf, p)
                v = buf[p:p + n]; p += n
            yield num, wire, v

    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        assert struct.unpack("<I", data[pos + 8:pos + 12])[0] == masked_crc(head), "length CRC"
        payload = data[pos + 12:pos + 12 + n]
        assert struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] == masked_crc(payload), "payload CRC"
        pos += 16 + n
        step, scal = 0, {}
        for num, wire, v in fields(payload):
            if num == 2:
                step = v
            elif num == 5:
                for n1, _, val in fields(v):
                    tag, x = None, None
                    for n2, _, vv in fields(val):
                        if n2 == 1:
                            tag = vv.decode()
                        elif n2 == 2:
                            (x,) = struct.unpack("<f", vv)
                    scal[tag] = x
        out.append((step, scal))
    return out
