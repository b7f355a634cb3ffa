"""The hand-rolled TensorBoard event writer: CRC32C known answers, TFRecord framing, protobuf round trip."""
import struct

from pioneer_amd import tb


def test_crc32c_known_answers():
    # RFC 3720 B.4 / the CRC-32C check value
    assert tb.crc32c(b"123456789") == 0xE3069283
    assert tb.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert tb.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E


def test_masking_matches_the_tfrecord_rule():
    c = tb.crc32c(b"abc")
    assert tb.masked_crc(b"abc") == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_event_file_round_trip(tmp_path):
    w = tb.ScalarWriter(str(tmp_path))
    w.add_scalars({"episode_reward_mean": 12.5, "episodes_total": 7, "trial_id": "00000", "nan_value": float("nan"),
                   "flag": True}, step=3)
    w.add_scalars({"episode_reward_mean": -1.25}, step=4)
    w.close()
    ev = tb.read_events(w.path)
    assert len(ev) == 3 and ev[0] == (0, {})                              # the file_version header event
    assert ev[1][0] == 3 and ev[1][1] == {"ray/tune/episode_reward_mean": 12.5, "ray/tune/episodes_total": 7.0}
    assert ev[2] == (4, {"ray/tune/episode_reward_mean": -1.25})
    raw = open(w.path, "rb").read()
    (n,) = struct.unpack("<Q", raw[:8])
    assert raw[12:12 + n].endswith(b"brain.Event:2")                      # TensorBoard's file-version marker
