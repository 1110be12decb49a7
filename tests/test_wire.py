"""Placement -> wire: the adapter writes crane.grpc.ResourceInNodeV3 / JobToD bytes straight from the
packed placements.  Here protobuf itself is the checker: the message classes are built at run time from descriptors that
restate /root/reference/protos/PublicDefs.proto:33-44 (Slots, DeviceTypeSlotsMap, DedicatedResourceInNode), :63-69
(ResourceInNodeV3) and :396-409 (JobToD, the fields the scheduler fills), the adapter's bytes are parsed with them, the
parsed message is compared field by field with the ResourceInNodeV3 OBJECT the write-back builds from the same packed
record, and the bytes are compared with protobuf's own deterministic serialisation of that message.  Host-only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cranesched_amd", "host", "test_host_adapter")


def _messages():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="crane_wire_test.proto", package="crane.grpc.t", syntax="proto3")

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, number, ftype, label
        if type_name:
            f.type_name = type_name
        return f

    def map_field(m, name, number, value_type_name):
        e = m.nested_type.add()
        e.name = "".join(w.capitalize() for w in name.split("_")) + "Entry"
        e.options.map_entry = True
        field(e, "key", 1, F.TYPE_STRING)
        field(e, "value", 2, F.TYPE_MESSAGE, type_name=value_type_name)
        field(m, name, number, F.TYPE_MESSAGE, F.LABEL_REPEATED, f".crane.grpc.t.{m.name}.{e.name}")

    s = msg("Slots")
    field(s, "slots", 1, F.TYPE_STRING, F.LABEL_REPEATED)
    map_field(msg("DeviceTypeSlotsMap"), "type_slots_map", 1, ".crane.grpc.t.Slots")
    map_field(msg("DedicatedResourceInNode"), "name_type_map", 1, ".crane.grpc.t.DeviceTypeSlotsMap")
    r = msg("ResourceInNodeV3")
    field(r, "cpu_ids", 1, F.TYPE_UINT32, F.LABEL_REPEATED)
    field(r, "cpu_count", 2, F.TYPE_DOUBLE)
    field(r, "memory_bytes", 3, F.TYPE_UINT64)
    field(r, "memory_sw_bytes", 4, F.TYPE_UINT64)
    field(r, "gres", 5, F.TYPE_MESSAGE, type_name=".crane.grpc.t.DedicatedResourceInNode")
    j = msg("JobToD")
    field(j, "job_id", 1, F.TYPE_UINT32)
    field(j, "uid", 2, F.TYPE_UINT32)
    field(j, "res", 4, F.TYPE_MESSAGE, type_name=".crane.grpc.t.ResourceInNodeV3")
    field(j, "partition", 5, F.TYPE_STRING)
    field(j, "account", 6, F.TYPE_STRING)
    field(j, "qos", 7, F.TYPE_STRING)
    field(j, "name", 9, F.TYPE_STRING)
    at = msg("ArrayTaskIdentity")                      # PublicDefs.proto:161-164
    field(at, "array_job_id", 1, F.TYPE_UINT32)
    field(at, "task_id", 2, F.TYPE_UINT32)
    f16 = field(j, "array_task", 16, F.TYPE_MESSAGE, type_name=".crane.grpc.t.ArrayTaskIdentity")   # `optional ArrayTaskIdentity array_task = 16`
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return {n: get(pool.FindMessageTypeByName(f"crane.grpc.t.{n}")) for n in ("ResourceInNodeV3", "JobToD")}


def _records(path):
    rec = None
    for line in open(path):
        tag, _, rest = line.rstrip("\n").partition(" ")
        if tag == "REC":
            rec = {"len": int(rest), "gres": {}, "extra": {}}
        elif tag == "HEX":
            rec["wire"] = bytes.fromhex(rest)
        elif tag == "CPU":
            rec["cpu"] = float(rest)
        elif tag == "MEM":
            rec["mem"], rec["mem_sw"] = (int(x) for x in rest.split())
        elif tag == "IDS":
            rec["ids"] = [int(x) for x in rest.split()]
        elif tag == "GRES":
            name, typ, *slots = rest.split()
            rec["gres"].setdefault(name, {})[typ] = slots
        elif tag in ("JOB", "JOBHEX", "ARRAY"):
            rec["extra"][tag] = rest
        elif tag == "END":
            yield rec


def test_resource_wire_parses_to_the_written_back_object(built, tmp_path):
    pytest.importorskip("google.protobuf")
    M = _messages()
    out = tmp_path / "wire.txt"
    r = subprocess.run([EXE, "--wire-dump", str(out), "900"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    n = 0
    kinds = set()
    for rec in _records(out):
        assert len(rec["wire"]) == rec["len"]
        m = M["ResourceInNodeV3"]()
        m.ParseFromString(rec["wire"])
        assert list(m.cpu_ids) == rec["ids"] == sorted(rec["ids"])
        assert m.cpu_count == rec["cpu"]
        assert m.memory_bytes == rec["mem"] and m.memory_sw_bytes == rec["mem_sw"]
        assert m.HasField("gres")                       # mutable_gres(): always present (PublicHeader.cpp:994)
        got = {name: {t: list(s.slots) for t, s in tm.type_slots_map.items()} for name, tm in m.gres.name_type_map.items()}
        assert got == rec["gres"]
        for tm in got.values():
            for slots in tm.values():
                assert slots == sorted(slots) and slots   # std::set order, no empty Slots entry
        # the very bytes protobuf writes for this message (deterministic = map keys sorted, as std::map iterates)
        assert m.SerializeToString(deterministic=True) == rec["wire"]
        kinds.add((bool(rec["ids"]), bool(rec["gres"]), rec["cpu"] == 0))
        n += 1
    assert n == 900 and len(kinds) >= 4


def test_job_to_d_wire(built, tmp_path):
    pytest.importorskip("google.protobuf")
    M = _messages()
    out = tmp_path / "jobtod.txt"
    r = subprocess.run([EXE, "--wire-dump", str(out), "60", "jobtod"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    n = children = 0
    for rec in _records(out):
        if "JOBHEX" not in rec["extra"]:
            continue
        job_id, uid, part, acct, qos, name = rec["extra"]["JOB"].split(" ")
        m = M["JobToD"]()
        raw = bytes.fromhex(rec["extra"]["JOBHEX"])
        m.ParseFromString(raw)
        assert (m.job_id, m.uid) == (int(job_id), int(uid))
        undash = lambda v: "" if v == "-" else v
        assert (m.partition, m.account, m.qos, m.name) == (undash(part), acct, undash(qos), undash(name))
        assert m.res.SerializeToString(deterministic=True) == rec["wire"]
        assert m.SerializeToString(deterministic=True) == raw
        if "ARRAY" in rec["extra"]:        # array children carry their identity (CtldPublicDefs.cpp:547-551), others no field 16 at all
            aj, at = map(int, rec["extra"]["ARRAY"].split(" "))
            assert m.HasField("array_task") and (m.array_task.array_job_id, m.array_task.task_id) == (aj, at)
            children += 1
        else:
            assert not m.HasField("array_task")
        n += 1
    assert n == 60 and children == 20
