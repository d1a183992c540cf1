"""Hand-assembled wire-format fixtures for the replay harness's readers.

Deliberately independent of eskf_lio_amd.replay: nothing here imports the package, and no byte is produced by its
writer.  Every message is laid out by hand from the public definitions —
  * OMG CDR (DDS-XTypes 1.3, 7.4.3 "plain CDR version 1" as ROS 2 / rmw serialises it): a 4-byte encapsulation header
    {0x00, 0x01 = CDR little endian | 0x00 = big endian, options 0x0000}; every primitive aligned to its own size
    COUNTED FROM THE BYTE AFTER THAT HEADER; string = uint32 length (terminating NUL included) + bytes + NUL;
    sequence = uint32 count + elements; fixed array = elements only;
  * std_msgs/Header {builtin_interfaces/Time stamp {int32 sec, uint32 nanosec}, string frame_id};
  * sensor_msgs/Imu {Header, geometry_msgs/Quaternion orientation (x y z w float64), float64[9] orientation_covariance,
    Vector3 angular_velocity, float64[9], Vector3 linear_acceleration, float64[9]};
  * sensor_msgs/PointField {string name, uint32 offset, uint8 datatype, uint32 count} with FLOAT32 = 7, FLOAT64 = 8,
    UINT16 = 4; sensor_msgs/PointCloud2 {Header, uint32 height, uint32 width, PointField[] fields, bool is_bigendian,
    uint32 point_step, uint32 row_step, uint8[] data, bool is_dense};
  * the rosbag2 sqlite3 storage plugin's schema (tables `topics` and `messages`, index on messages.timestamp, and the
    `schema` / `metadata` tables newer versions add), written with raw SQL —
with the offsets asserted as the bytes are appended.  What the reference does with these messages:
/root/reference/include/ESKF_LIO/Subscriber.hpp:38-52 (Imu) and :80-103 (PointCloud2: float x y z, double `timestamp`).

Outputs (committed): imu_le.cdr, imu_be.cdr, cloud_le.cdr, cloud_be.cdr, cloud_bad_datatype.cdr, cloud_pl_cdr.cdr,
mini_bag.db3 and wire_expected.json.   usage: python tests/golden/make_wire_fixtures.py
"""
import json
import os
import sqlite3
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


class Stream:
    """Bytes after the encapsulation header; align() is relative to the start of THIS buffer."""

    def __init__(self, little=True):
        self.b = bytearray()
        self.e = "<" if little else ">"

    def align(self, n):
        while len(self.b) % n:
            self.b.append(0)

    def put(self, fmt, *v):
        self.align(struct.calcsize(fmt[0]) if len(fmt) == 1 else struct.calcsize(fmt[-1]))
        self.b += struct.pack(self.e + fmt, *v)

    def string(self, s):
        raw = s.encode("ascii") + b"\0"
        self.put("I", len(raw))
        self.b += raw


def header(st, sec, nanosec, frame_id):
    st.put("i", sec)
    st.put("I", nanosec)
    st.string(frame_id)


def imu(little):
    st = Stream(little)
    header(st, 1_646_000_123, 456_789_012, "imu_sensor_frame")            # 16 chars: the string ends at offset 29
    assert len(st.b) == 4 + 4 + 4 + 17
    st.align(8)
    assert len(st.b) == 32                                                # three bytes of padding before the first float64
    for v in (0.0, 0.0, 0.38268343236508978, 0.92387953251128674):        # orientation x y z w (unused by the reference)
        st.put("d", v)
    for k in range(9):
        st.put("d", 0.01 * k)                                             # orientation_covariance
    for v in (0.011, -0.022, 0.033):                                      # angular_velocity
        st.put("d", v)
    for k in range(9):
        st.put("d", -1.0)                                                 # "covariance unknown"
    for v in (0.15, -0.25, 9.80665):                                      # linear_acceleration
        st.put("d", v)
    for k in range(9):
        st.put("d", 0.0)
    assert len(st.b) == 32 + 8 * (4 + 9 + 3 + 9 + 3 + 9)
    return bytes([0x00, 0x01 if little else 0x00, 0x00, 0x00]) + bytes(st.b)


POINTS = [  # x, y, z, intensity, ring, timestamp — 2 rows x 3 columns
    (1.5, -2.25, 0.125, 17.0, 3, 1646000123.000001),
    (-10.0, 20.5, -0.75, 0.5, 31, 1646000123.016667),
    (0.1, 0.2, 0.3, 255.0, 0, 1646000123.033334),      # 0.1 is not a float32: the widening to double must show it
    (100.0, -100.0, 3.0, 1.0, 7, 1646000123.050001),
    (0.0, 0.0, 0.0, 0.0, 1, 1646000123.066668),
    (-0.5, 4.75, 12.0, 42.0, 15, 1646000123.099999),
]


def cloud(little, x_datatype=7, encapsulation=None):
    st = Stream(little)
    header(st, 1_646_000_123, 99_999_000, "PandarXT-32")                 # 11 chars: ends at 8 + 4 + 12 = 24, aligned already
    assert len(st.b) == 24
    st.put("I", 2)                                                        # height
    st.put("I", 3)                                                        # width
    # point layout (32 bytes): x f32 @0, y f32 @4, z f32 @8, intensity f32 @12, ring u16 @16, 6 bytes of padding,
    # timestamp f64 @24 — two fields the reference does not read sit BEFORE `timestamp`, and the step includes padding
    fields = [("x", 0, x_datatype), ("y", 4, 7), ("z", 8, 7), ("intensity", 12, 7), ("ring", 16, 4), ("timestamp", 24, 8)]
    st.put("I", len(fields))
    for name, offset, datatype in fields:
        st.string(name)
        st.put("I", offset)                                               # aligned to 4 after the string's NUL
        st.put("B", datatype)
        st.put("I", 1)                                                    # count: three bytes of padding after the uint8
    st.put("B", 0 if little else 1)                                       # is_bigendian follows the stream's byte order here
    st.put("I", 32)                                                       # point_step
    st.put("I", 96)                                                       # row_step
    st.put("I", 32 * len(POINTS))                                         # uint8[] data: count, then the bytes (no alignment)
    e = "<" if little else ">"
    for x, y, z, inten, ring, t in POINTS:
        rec = struct.pack(e + "ffffH", x, y, z, inten, ring) + b"\xAA" * 6 + struct.pack(e + "d", t)
        assert len(rec) == 32
        st.b += rec
    st.put("B", 1)                                                        # is_dense
    enc = encapsulation if encapsulation is not None else bytes([0x00, 0x01 if little else 0x00, 0x00, 0x00])
    return enc + bytes(st.b)


def f32(v):
    return struct.unpack("<f", struct.pack("<f", v))[0]


def main():
    blobs = {
        "imu_le.cdr": imu(True), "imu_be.cdr": imu(False),
        "cloud_le.cdr": cloud(True), "cloud_be.cdr": cloud(False),
        "cloud_bad_datatype.cdr": cloud(True, x_datatype=8),              # x declared FLOAT64: the reference's iterator<float> would misread
        "cloud_pl_cdr.cdr": cloud(True, encapsulation=bytes([0x00, 0x03, 0x00, 0x00])),   # PL_CDR_LE: not plain CDR
    }
    for name, b in blobs.items():
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(b)
    # ---- a minimal rosbag2 sqlite3 file, raw SQL, the plugin's schema ----
    path = os.path.join(HERE, "mini_bag.db3")
    if os.path.exists(path):
        os.remove(path)
    db = sqlite3.connect(path)
    db.executescript("""
        CREATE TABLE schema(schema_version INTEGER PRIMARY KEY, ros_distro TEXT NOT NULL);
        CREATE TABLE metadata(id INTEGER PRIMARY KEY, metadata_version INTEGER NOT NULL, metadata TEXT NOT NULL);
        CREATE TABLE topics(id INTEGER PRIMARY KEY, name TEXT NOT NULL, type TEXT NOT NULL, serialization_format TEXT NOT NULL,
                            offered_qos_profiles TEXT NOT NULL);
        CREATE TABLE messages(id INTEGER PRIMARY KEY, topic_id INTEGER NOT NULL, timestamp INTEGER NOT NULL, data BLOB NOT NULL);
        CREATE INDEX timestamp_idx ON messages (timestamp ASC);
        INSERT INTO schema VALUES (3, 'humble');
        INSERT INTO topics VALUES (7, '/tf_static', 'tf2_msgs/msg/TFMessage', 'cdr', '');
        INSERT INTO topics VALUES (3, '/alphasense/imu', 'sensor_msgs/msg/Imu', 'cdr', '');
        INSERT INTO topics VALUES (5, '/hesai/pandar', 'sensor_msgs/msg/PointCloud2', 'cdr', '');
    """)
    rows = [  # inserted OUT of bag-time order; a player publishes by timestamp
        (5, 1_646_000_123_200_000_000, blobs["cloud_le.cdr"]),
        (7, 1_646_000_123_000_000_000, b"\x00\x01\x00\x00" + b"\x00" * 8),      # a topic the reference does not subscribe to
        (3, 1_646_000_123_100_000_000, blobs["imu_le.cdr"]),
        (3, 1_646_000_123_300_000_000, blobs["imu_be.cdr"]),
        (5, 1_646_000_123_050_000_000, blobs["cloud_be.cdr"]),
    ]
    db.executemany("INSERT INTO messages(topic_id, timestamp, data) VALUES (?, ?, ?)", rows)
    db.commit()
    db.close()
    expected = {
        "imu": {"timestamp": 1_646_000_123 + 1e-9 * 456_789_012, "angular_velocity": [0.011, -0.022, 0.033],
                "linear_acceleration": [0.15, -0.25, 9.80665]},
        "cloud": {"points": [[f32(p[0]), f32(p[1]), f32(p[2])] for p in POINTS], "point_time": [p[5] for p in POINTS]},
        "bag_order": [["cloud", 1646000123.05], ["imu", 1646000123.1], ["cloud", 1646000123.2], ["imu", 1646000123.3]],
    }
    with open(os.path.join(HERE, "wire_expected.json"), "w") as f:
        json.dump(expected, f, indent=1)
    print({k: len(v) for k, v in blobs.items()}, os.path.getsize(path))


if __name__ == "__main__":
    main()
