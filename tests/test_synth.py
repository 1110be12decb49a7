"""The synthetic queues are frozen: seeds, distributions and sizes."""
import zlib

import numpy as np

from cranesched_amd import synth


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).view(np.uint8), c)
    return c


def test_splitmix64_reference_values():
    # splitmix64 from seed 0: first outputs of the public reference implementation
    out = synth.splitmix64(0, 3)
    assert [int(x) for x in out] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_configs_are_deterministic_and_sized():
    for name, (J, N, P) in {"C1": (1000, 128, 1), "C2": (100000, 4096, 1)}.items():
        c, j, now = synth.make_config(name)
        assert (j.num_jobs, c.num_nodes, c.num_partitions, now) == (J, N, P, synth.NOW)
        c2, j2, _ = synth.make_config(name)
        assert crc(j.time_limit_sec, j.task_cpu_raw, j.partition) == crc(j2.time_limit_sec, j2.task_cpu_raw, j2.partition)


def test_c4_shape():
    c, j, _ = synth.make_config("C4", J=50000)
    assert c.num_nodes == 65536 and c.num_partitions == 8
    k = j.node_num
    assert set(np.unique(k)) == {1, 2, 4, 8} and 0.88 < (k == 1).mean() < 0.92
    assert (j.ntasks == k).all()
    g = j.gres_total
    assert 0.18 < (g[:, 0] > 0).mean() < 0.22 and 0.08 < (g[:, 1] > 0).mean() < 0.12
    assert crc(j.time_limit_sec[:1000], j.task_cpu_raw[:1000], j.partition[:1000], k[:1000]) == crc(
        *[getattr(synth.make_config("C4", J=1000)[1], f) for f in ("time_limit_sec", "task_cpu_raw", "partition", "node_num")])
