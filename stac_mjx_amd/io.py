"""Data loading / saving with the reference's conventions.

Mirrors ``stac_mjx/io.py``: ``StacData`` (:17-36), ``load_data`` (:39-98, keypoints re-ordered to
``KEYPOINT_MODEL_PAIRS`` key order, scaled by ``MOCAP_SCALE_FACTOR``, flattened to
``[frames, 3K]`` as ``[kp0.x, kp0.y, kp0.z, kp1.x, ...]``), ``save_data_to_h5`` (:194-237) and
``load_stac_data`` (:240-278).

``.mat`` needs only scipy.  ``.h5``/``.nwb`` inputs and ``.h5`` outputs need h5py, which is
optional in this image: without it results are written as ``.npz`` holding the same dataset
names (the output path keeps its stem, the suffix becomes ``.npz``).
"""

from __future__ import annotations

from dataclasses import asdict, dataclass, field
from pathlib import Path

import numpy as np
import yaml

from .config import ConfigNode, validate_config

try:  # optional
    import h5py  # type: ignore
except Exception:  # pragma: no cover - depends on the image
    h5py = None


@dataclass
class StacData:
    """Output record (same fields as ``stac_mjx.io.StacData``)."""

    qpos: np.ndarray
    xpos: np.ndarray
    xquat: np.ndarray
    marker_sites: np.ndarray
    offsets: np.ndarray
    kp_data: np.ndarray
    names_qpos: list
    names_xpos: list
    kp_names: list
    qvel: np.ndarray = field(default_factory=lambda: np.array([]))

    def as_dict(self) -> dict:
        return asdict(self)


def load_dannce(filename, names_filename=None):
    """``.mat`` loader (io.py:101-124): returns ``pred`` as stored, i.e. [frames, xyz, keypoints]."""
    import scipy.io as spio

    node_names = None
    if names_filename is not None:
        mat = spio.loadmat(names_filename)
        node_names = [item[0] for sublist in mat["joint_names"] for item in sublist]
    data = spio.loadmat(filename, struct_as_record=False, squeeze_me=True)["pred"]
    return np.asarray(data), node_names


def load_h5(filename):
    """``.h5`` loader (io.py:150-170): dataset ``tracks`` [frames, 1, keypoints, xyz] -> [frames, xyz, keypoints]."""
    if h5py is None:
        raise ImportError("h5py is required to read .h5 mocap files")
    with h5py.File(filename, "r") as f:
        data = np.array(f["tracks"][()])
    data = np.squeeze(data, axis=1)
    return np.transpose(data, (0, 2, 1)), None


def load_nwb(filename):
    """``.nwb`` loader (io.py:127-147) read through plain h5py (pynwb/ndx_pose are not needed for the
    PoseEstimationSeries layout): returns [frames, xyz, keypoints] and the node names."""
    if h5py is None:
        raise ImportError("h5py is required to read .nwb mocap files")
    with h5py.File(filename, "r") as f:
        pe = f["processing/behavior/PoseEstimation"]
        if "nodes" in pe:
            node_names = [n.decode() if isinstance(n, bytes) else str(n) for n in pe["nodes"][()]]
        else:
            node_names = [k for k in pe.keys() if isinstance(pe[k], h5py.Group) and "data" in pe[k]]
        data = np.stack([pe[n]["data"][()] for n in node_names], axis=-1)
    return data, node_names


def load_data(cfg, base_path: Path | None = None):
    """Load, re-order, scale and flatten mocap data (io.py:39-98).

    Returns ``(kp_data [frames, 3K] float32, sorted_kp_names)``.
    """
    base_path = Path.cwd() if base_path is None else Path(base_path)
    file_path = base_path / cfg.stac.data_path
    if file_path.suffix == ".mat":
        label3d_path = cfg.model.get("KP_NAMES_LABEL3D_PATH", None)
        data, kp_names = load_dannce(str(file_path), names_filename=label3d_path)
    elif file_path.suffix == ".nwb":
        data, kp_names = load_nwb(file_path)
    elif file_path.suffix == ".h5":
        data, kp_names = load_h5(file_path)
    else:
        raise ValueError("Unsupported file extension. Please provide a .mat, .nwb, or .h5 file.")

    kp_names = kp_names or cfg.model.get("KP_NAMES", None)
    if kp_names is None:
        raise ValueError(
            "Keypoint names not provided. Please provide an ordered list of keypoint names "
            "corresponding to the keypoint data order."
        )
    kp_names = list(kp_names)
    if len(kp_names) != data.shape[2]:
        raise ValueError(
            f"Number of keypoint names ({len(kp_names)}) is not the same as the number of keypoints in data ({data.shape[2]})"
        )
    model_inds = [kp_names.index(src) for src in cfg.model.KEYPOINT_MODEL_PAIRS.keys()]
    sorted_kp_names = [kp_names[i] for i in model_inds]
    data = np.asarray(data, dtype=np.float64) * cfg.model.MOCAP_SCALE_FACTOR  # float64, then cast (io.py:93-94)
    data = data[:, :, model_inds].astype(np.float32)
    data = np.transpose(data, (0, 2, 1)).reshape(data.shape[0], -1)
    return np.ascontiguousarray(data), sorted_kp_names


_DATASETS = ("kp_data", "marker_sites", "offsets", "qpos", "qvel", "xpos", "xquat")


def resolve_output_path(file_path) -> Path:
    """The path ``save_data_to_h5`` writes for ``file_path``: itself with h5py, else its ``.npz`` stand-in."""
    file_path = Path(file_path)
    if h5py is not None and file_path.suffix in (".h5", ".hdf5"):
        return file_path
    return file_path.with_suffix(".npz")


# ---- result files: the reference's format, compressed on every host core -------------------------------------------------------
# `io.py:224-236` fixes the container and the filter (h5py datasets with compression="gzip"), not how the deflate streams are
# produced.  h5py deflates chunk after chunk on ONE thread: 5.4 s for the 273 MB of a 100 000-frame rodent run, seven times the
# GPU time of the run itself.  Here the chunks are deflated by a thread pool (zlib releases the GIL) and handed
# to HDF5 ready-made (`write_direct_chunk`: the same filter pipeline as an h5py write -- any reader inflates them --; the chunks are
# row blocks of about 1 MiB, not h5py's auto-chunk guess, which readers do not see; at most a window of compressed chunks is in memory
# at a time, and the ranks of a node divide its cores among themselves); the `.npz` stand-in of an interpreter without h5py gets the same treatment (its members are ordinary
# deflate streams put together pigz-style from independently compressed blocks).
_PAR_MIN_BYTES = 1 << 20   # arrays below this are written the plain way
_NPZ_BLOCK = 1 << 20       # bytes of input per deflate block of an .npz member (a block is 40 ms of one core at level 6)
_H5_GZIP_LEVEL = 4         # h5py's default for compression="gzip" (the reference passes no level)


def _n_threads() -> int:
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    # one process per GPU: the ranks of a node share its cores (per-rank shard files are written by every rank at once)
    try:
        local_world = max(int(os.environ.get("LOCAL_WORLD_SIZE", "1")), 1)
    except ValueError:
        local_world = 1
    # (measured on a 256-thread host, 273 MB: 16 threads 0.35 s, 32 0.28 s, 64 0.31 s, 96 0.30 s, 128 0.36-0.39 s -- from 32 threads on the
    #  serial side, the one thread that cuts the jobs and writes the file, is what is left; STAC_IO_THREADS overrides)
    try:
        forced = int(os.environ.get("STAC_IO_THREADS", "0"))
    except ValueError:
        forced = 0
    return forced if forced > 0 else max(1, min(n // local_world, 64))


def _map_window(pool, fn, jobs, window=None):
    """``pool.map`` in order with a bounded number of jobs in flight (results are consumed -- written -- as they come, so at most
    ``window`` compressed chunks exist at a time instead of the whole dataset's)."""
    from collections import deque

    window = window or 4 * _n_threads()
    pending = deque()
    it = iter(jobs)
    for job in it:
        pending.append(pool.submit(fn, job))
        if len(pending) >= window:
            yield pending.popleft().result()
    while pending:
        yield pending.popleft().result()


def _pool():
    from concurrent.futures import ThreadPoolExecutor

    return ThreadPoolExecutor(max_workers=_n_threads())


def _h5_write_parallel(f, name, arr, pool):
    """One gzip dataset, its chunks deflated in parallel.  Chunks are whole rows (leading axis), about 1 MiB each."""
    import zlib

    arr = np.ascontiguousarray(arr)
    row_bytes = max(arr.dtype.itemsize * int(np.prod(arr.shape[1:], dtype=np.int64)), 1)
    rows = int(max(1, min(arr.shape[0], (1 << 20) // row_bytes)))
    d = f.create_dataset(name, shape=arr.shape, dtype=arr.dtype, chunks=(rows,) + tuple(arr.shape[1:]), compression="gzip",
                         compression_opts=_H5_GZIP_LEVEL)
    n_chunks = (arr.shape[0] + rows - 1) // rows

    def deflate(i):
        blk = arr[i * rows:(i + 1) * rows]
        if blk.shape[0] < rows:  # HDF5 stores whole chunks: the last one is padded (the padding is never read back)
            pad = np.zeros((rows,) + arr.shape[1:], arr.dtype)
            pad[:blk.shape[0]] = blk
            blk = pad
        return zlib.compress(memoryview(blk).cast("B"), _H5_GZIP_LEVEL)

    zeros = (0,) * (arr.ndim - 1)
    for i, comp in enumerate(_map_window(pool, deflate, range(n_chunks))):  # (in order; HDF5 itself is single-threaded)
        d.id.write_direct_chunk((i * rows,) + zeros, comp)


def _npy_bytes_header(arr) -> bytes:
    import io as _io

    b = _io.BytesIO()
    np.lib.format.write_array_header_1_0(b, np.lib.format.header_data_from_array_1_0(arr))
    return b.getvalue()


_CRC_SHIFT = {}  # len2 -> the 32 x 32 GF(2) matrix (32 column words) that appends len2 zero bytes to a CRC-32 register


def _crc32_shift_matrix(len2: int):
    """Appending zero bytes to a message is a linear map of its CRC register over GF(2); the map for len2 bytes by repeated squaring
    of the one-bit map (cached per length: every deflate block but a member's last has the same)."""
    mat = _CRC_SHIFT.get(len2)
    if mat is not None:
        return mat

    def times(m, vec):
        out, i = 0, 0
        while vec:
            if vec & 1:
                out ^= m[i]
            vec >>= 1
            i += 1
        return out

    def mul(m1, m2):  # m1 after m2
        return [times(m1, m2[i]) for i in range(32)]

    one_bit = [0xEDB88320] + [1 << (i - 1) for i in range(1, 32)]  # one zero BIT appended (reflected polynomial)
    power = one_bit
    for _ in range(3):
        power = mul(power, power)  # eight bits = one zero byte
    result = [1 << i for i in range(32)]  # identity
    n = len2
    while n:
        if n & 1:
            result = mul(power, result)
        n >>= 1
        if n:
            power = mul(power, power)
    if len(_CRC_SHIFT) < 64:
        _CRC_SHIFT[len2] = result
    return result


def _crc32_combine(crc1: int, crc2: int, len2: int) -> int:
    """CRC-32 of A + B from crc32(A), crc32(B) and len(B): lets every deflate block carry its own CRC."""
    if len2 <= 0:
        return crc1
    mat = _crc32_shift_matrix(len2)
    out, i = 0, 0
    while crc1:
        if crc1 & 1:
            out ^= mat[i]
        crc1 >>= 1
        i += 1
    return (out ^ crc2) & 0xFFFFFFFF


def _npz_write_parallel(path, members: dict, pool, level: int = _H5_GZIP_LEVEL) -> None:
    """``np.savez_compressed`` with the deflate work spread over the pool: every member is a plain ZIP_DEFLATED entry whose
    stream is the concatenation of independently compressed blocks (each ended with a sync flush, the last one finished), so
    ``np.load`` / ``zipfile`` read it like any other .npz.  ZIP64 throughout (members can exceed 4 GiB).  The blocks of ALL members
    form one job stream (no member waits for the tail of the one before it), each block carries its own CRC (combined per
    member: _crc32_combine), and at most a window of compressed blocks is in memory at a time."""
    import struct
    import zlib

    def deflate(job):
        buf, last = job
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        return c.compress(buf) + c.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH), zlib.crc32(buf), len(buf)

    plan, jobs = [], []  # per member: (file name, blocks); the job stream over all of them
    for name, arr in members.items():
        arr = np.asarray(arr)
        if arr.dtype.hasobject:
            raise ValueError("object arrays are not written")
        arr = np.ascontiguousarray(arr)
        raw = memoryview(arr.reshape(-1)).cast("B") if arr.size else memoryview(b"")
        head = _npy_bytes_header(arr)
        blocks = [bytes(head) + bytes(raw[:max(_NPZ_BLOCK - len(head), 0)])]
        pos = max(_NPZ_BLOCK - len(head), 0)
        while pos < len(raw):
            blocks.append(raw[pos:pos + _NPZ_BLOCK])
            pos += _NPZ_BLOCK
        plan.append(((name + ".npy").encode(), len(blocks)))
        jobs.extend((b, i == len(blocks) - 1) for i, b in enumerate(blocks))
    results = _map_window(pool, deflate, jobs)
    central = []
    with open(path, "wb") as fh:
        for fname, n_blocks in plan:
            offset = fh.tell()
            # local header: version 45 (ZIP64), no flags, deflate, DOS time 1980-01-01; CRC and sizes are patched in once the member's
            # blocks have been written
            fh.write(struct.pack("<IHHHHHIIIHH", 0x04034B50, 45, 0, 8, 0, 0x21, 0, 0xFFFFFFFF, 0xFFFFFFFF, len(fname), 20))
            fh.write(fname)
            extra_at = fh.tell()
            fh.write(struct.pack("<HHQQ", 1, 16, 0, 0))
            crc = csize = usize = 0
            for _ in range(n_blocks):
                comp, bcrc, blen = next(results)
                fh.write(comp)
                crc = _crc32_combine(crc, bcrc, blen) if usize else bcrc
                csize += len(comp)
                usize += blen
            crc &= 0xFFFFFFFF
            end = fh.tell()
            fh.seek(offset + 14)
            fh.write(struct.pack("<I", crc))
            fh.seek(extra_at)
            fh.write(struct.pack("<HHQQ", 1, 16, usize, csize))
            fh.seek(end)
            central.append((fname, crc, csize, usize, offset))
        cd_start = fh.tell()
        for fname, crc, csize, usize, offset in central:
            extra = struct.pack("<HHQQQ", 1, 24, usize, csize, offset)
            fh.write(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 45, 45, 0, 8, 0, 0x21, crc, 0xFFFFFFFF, 0xFFFFFFFF, len(fname),
                                 len(extra), 0, 0, 0, 0o600 << 16, 0xFFFFFFFF))
            fh.write(fname)
            fh.write(extra)
        cd_size = fh.tell() - cd_start
        eocd64 = fh.tell()
        fh.write(struct.pack("<IQHHIIQQQQ", 0x06064B50, 44, 45, 45, 0, 0, len(central), len(central), cd_size, cd_start))
        fh.write(struct.pack("<IIQI", 0x07064B50, 0, eocd64, 1))
        n16 = min(len(central), 0xFFFF)
        fh.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, n16, n16, 0xFFFFFFFF, 0xFFFFFFFF, 0))


def save_data_to_h5(config, kp_names, names_qpos, names_xpos, kp_data, marker_sites, offsets, qpos,
                    xpos, xquat, qvel, file_path) -> Path:  # fmt: skip
    """Write the reference's output contract (io.py:194-237).  Returns the path actually written.  Same datasets, dtypes
    and gzip filter as the reference; the deflate work of the large arrays runs on every host core (see above)."""
    file_path = Path(file_path)
    cfg_yaml = config.to_yaml() if isinstance(config, ConfigNode) else yaml.safe_dump(config, sort_keys=False)
    arrays = dict(kp_data=kp_data, marker_sites=marker_sites, offsets=offsets, qpos=qpos,
                  qvel=np.asarray(qvel), xpos=xpos, xquat=xquat)  # fmt: skip
    if h5py is not None and file_path.suffix in (".h5", ".hdf5"):
        with h5py.File(file_path, "w") as f, _pool() as pool:
            f.create_dataset("config", data=np.bytes_(cfg_yaml))
            f.create_dataset("kp_names", data=np.array(kp_names, dtype="S"))
            f.create_dataset("names_qpos", data=np.array(names_qpos, dtype="S"))
            f.create_dataset("names_xpos", data=np.array(names_xpos, dtype="S"))
            for k, v in arrays.items():
                v = np.asarray(v)
                if v.ndim and v.nbytes >= _PAR_MIN_BYTES and hasattr(h5py.h5d.DatasetID, "write_direct_chunk"):
                    _h5_write_parallel(f, k, v, pool)
                else:
                    f.create_dataset(k, data=v, compression="gzip" if v.ndim else None)
        return file_path
    out = file_path.with_suffix(".npz")
    members = dict(config=np.bytes_(cfg_yaml), kp_names=np.array(kp_names, dtype="S"), names_qpos=np.array(names_qpos, dtype="S"),
                   names_xpos=np.array(names_xpos, dtype="S"), **{k: np.asarray(v) for k, v in arrays.items()})
    with _pool() as pool:
        _npz_write_parallel(out, members, pool)
    return out


def load_stac_data(file_path):
    """Read back a result file (io.py:240-278).  Accepts the ``.h5`` path or its ``.npz`` stand-in."""
    file_path = Path(file_path)
    if not file_path.exists() and file_path.with_suffix(".npz").exists():
        file_path = file_path.with_suffix(".npz")
    if file_path.suffix == ".npz":
        with np.load(file_path, allow_pickle=False) as f:
            d = {k: f[k] for k in f.files}
        cfg_yaml = bytes(d["config"]).decode("utf-8")
    else:
        if h5py is None:
            raise ImportError("h5py is required to read .h5 result files")
        with h5py.File(file_path, "r") as f:
            d = {k: f[k][()] for k in f.keys()}
        cfg_yaml = d["config"].decode("utf-8")
    config = validate_config(yaml.safe_load(cfg_yaml))

    def names(a):
        return [n.decode("utf-8") if isinstance(n, bytes) else str(n) for n in np.asarray(a).tolist()]

    return config, StacData(
        kp_names=names(d["kp_names"]), names_qpos=names(d["names_qpos"]), names_xpos=names(d["names_xpos"]),
        kp_data=d["kp_data"], marker_sites=d["marker_sites"], offsets=d["offsets"], qpos=d["qpos"],
        qvel=d["qvel"], xpos=d["xpos"], xquat=d["xquat"])  # fmt: skip


# ---- per-rank shard files of a multi-GPU ik_only run (engine extension; main.run_stac with stac.gather = none / auto) ----
def shard_path(file_path, rank: int, world_size: int) -> Path:
    """``ik_only.h5`` -> ``ik_only.rank003-of-008.h5``."""
    file_path = Path(file_path)
    return file_path.with_name(f"{file_path.stem}.rank{rank:03d}-of-{world_size:03d}{file_path.suffix}")


def manifest_path(file_path) -> Path:
    file_path = Path(file_path)
    return file_path.with_name(file_path.stem + ".manifest.json")


def write_manifest(manifest, file_path, world_size: int, n_clips: int, n_frames_per_clip: int) -> Path:
    """Which rank's file holds which clips (contiguous blocks, ``dist.shard_range``)."""
    import json

    from .dist import shard_range

    shards = []
    for r in range(world_size):
        lo, hi = shard_range(n_clips, r, world_size)
        shards.append(dict(rank=r, file=resolve_output_path(shard_path(file_path, r, world_size)).name, clip_lo=lo, clip_hi=hi,
                           frame_lo=lo * n_frames_per_clip, frame_hi=hi * n_frames_per_clip))
    doc = dict(format="stac_mjx_amd.sharded_ik_only/1", world_size=world_size, n_clips=n_clips,
               n_frames_per_clip=n_frames_per_clip, shards=shards)
    Path(manifest).write_text(json.dumps(doc, indent=1) + "\n")
    return Path(manifest)


def load_sharded_stac_data(manifest):
    """Read a sharded run back as ONE ``StacData`` (frame order = the order of an unsharded run)."""
    import json

    manifest = Path(manifest)
    doc = json.loads(manifest.read_text())
    cfg, parts = None, []
    for sh in doc["shards"]:
        cfg, d = load_stac_data(manifest.parent / sh["file"])
        if d.qpos.shape[0] != sh["frame_hi"] - sh["frame_lo"]:
            raise ValueError(f"shard {sh['file']} holds {d.qpos.shape[0]} frames, the manifest says {sh['frame_hi'] - sh['frame_lo']}")
        parts.append(d)
    cat = lambda name: np.concatenate([getattr(p, name) for p in parts if getattr(p, name).size], axis=0) if any(
        getattr(p, name).size for p in parts) else np.array([])  # noqa: E731
    first = parts[0]
    return cfg, StacData(qpos=cat("qpos"), xpos=cat("xpos"), xquat=cat("xquat"), marker_sites=cat("marker_sites"),
                         offsets=first.offsets, kp_data=cat("kp_data"), names_qpos=first.names_qpos,
                         names_xpos=first.names_xpos, kp_names=first.kp_names, qvel=cat("qvel"))
