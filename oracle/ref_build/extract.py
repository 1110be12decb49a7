#!/usr/bin/env python3
"""Slice the node-selection path out of the REFERENCE's own sources, at build time.

ORACLE / TEST INFRASTRUCTURE ONLY.  Reads /root/reference (read-only), writes generated
include files under oracle/_ref/gen/ — a directory that is git-ignored: reference text is
never committed to this repository.  oracle/Makefile (`make -C oracle _ref`) compiles the
generated files together with the committed stand-ins under oracle/ref_build/shim/ and the
committed driver oracle/ref_build/ref_harness.cpp into oracle/_ref/libcrane_ref*.so.

What is sliced:

  PublicHeader.h   the resource classes: TypeSlotsMap … ResourceView and the free operators
                   ("Public definitions for all components" up to IsFinishedStepStatus)
  PublicHeader.cpp every top-level definition whose signature does not mention a protobuf
                   type (crane::grpc::…): the whole resource algebra
  JobScheduler.h   namespace Ctld from its opening to the end of class SchedulerAlgo:
                   MinCpuTimeRatioFirst, RnJobInScheduler, PdJobInScheduler, BasicPriority,
                   MultiFactorPriority, SchedulerAlgo with NodeState, NodeSelector,
                   LocalScheduler, EarliestStartSubsetSelector, PreemptSegTree
  JobScheduler.cpp LocalScheduler::{CalculateRunningNodesAndStartTime_, GetNodesAndTrySchedule_,
                   Backfill_, TryPreempt_}, SchedulerAlgo::NodeSelect, MultiFactorPriority::*
  AccountMetaContainer.h    struct MetaResource … the end of class AccountMetaContainer
  AccountMetaContainer.cpp  the run-limit admission of the commit loop: MetaResource::operator+=,
                   AccountMetaContainer::{CheckAndMallocMetaResource, CheckTres_, IsUnlimitedTres_,
                   CheckQosRunLimitsForEntity_, CheckPartitionRunLimitsForEntity_, CheckEntityRunLimits_,
                   CheckRunLimits_, CheckGres_, LockAccountStripes_, DoMallocResource_}
  CtldPublicDefs.cpp        JobInCtld::SchedulePendingSteps
  AccountDefs.h             struct License
  LicenseManager.cpp        LicenseManager::CheckLicenseCountSufficient — the license pre-pass NodeSelect runs between ordering and
                   selection (JobScheduler.cpp:6739)

Every slice is located by ANCHOR TEXT (a changed reference fails loudly) and the line
ranges found are written to oracle/_ref/gen/MANIFEST.txt.  No line of a slice is edited.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import re
import sys

REF = "/root/reference"
PH_H = "src/Utilities/PublicHeader/include/crane/PublicHeader.h"
PH_CPP = "src/Utilities/PublicHeader/PublicHeader.cpp"
JS_H = "src/CraneCtld/JobScheduler.h"
JS_CPP = "src/CraneCtld/JobScheduler.cpp"
AMC_H = "src/CraneCtld/Accounting/AccountMetaContainer.h"
AMC_CPP = "src/CraneCtld/Accounting/AccountMetaContainer.cpp"
CPD_CPP = "src/CraneCtld/CtldPublicDefs.cpp"
ACD_H = "src/CraneCtld/Account/AccountDefs.h"
LM_CPP = "src/CraneCtld/Accounting/LicenseManager.cpp"


def read(rel):
    with open(os.path.join(REF, rel), encoding="utf-8") as f:
        return f.read().split("\n")


def find_line(lines, text, start=0, exact=False):
    for i in range(start, len(lines)):
        if (lines[i] == text) if exact else lines[i].startswith(text):
            return i
    raise SystemExit(f"extract.py: anchor not found: {text!r}")


def _code_of(ln, state):
    """The line without string literals and comments (enough for brace counting in these files)."""
    code = re.sub(r'"(\\.|[^"\\])*"', '""', ln)
    code = re.sub(r"'(\\.|[^'\\])'", "''", code)
    if state["block"]:
        if "*/" not in code:
            return ""
        code = code.split("*/", 1)[1]
        state["block"] = False
    code = re.sub(r"/\*.*?\*/", "", code)
    if "/*" in code:
        code = code.split("/*", 1)[0]
        state["block"] = True
    return code.split("//", 1)[0]


def top_level_defs(lines, first):
    """Split lines[first:] into top-level chunks: each ends where a definition's braces close
    (or at a `;` outside braces).  Comment / blank lines in front of a definition belong to it."""
    chunks, cur, depth, start, opened_any = [], [], 0, first, False
    state = {"block": False}
    for i in range(first, len(lines)):
        cur.append(lines[i])
        code = _code_of(lines[i], state)
        depth += code.count("{") - code.count("}")
        opened_any = opened_any or "{" in code
        if depth == 0 and ((opened_any and "}" in code) or (not opened_any and code.rstrip().endswith(";"))):
            chunks.append((start, i, cur))
            cur, start, opened_any = [], i + 1, False
    if any(c.strip() for c in cur):
        chunks.append((start, len(lines) - 1, cur))
    return chunks


def slice_block(lines, begin_anchor, end_anchor, include_end=False, start=0):
    b = find_line(lines, begin_anchor, start)
    e = find_line(lines, end_anchor, b + 1)
    if not include_end:
        e -= 1
    return b, e


def function_range(lines, sig_prefix, start=0):
    """[b, e] of the top-level function definition whose first line starts with sig_prefix."""
    b = find_line(lines, sig_prefix, start)
    depth, seen = 0, False
    for i in range(b, len(lines)):
        code = re.sub(r'"(\\.|[^"\\])*"', '""', lines[i]).split("//", 1)[0]
        depth += code.count("{") - code.count("}")
        if "{" in code:
            seen = True
        if seen and depth == 0:
            return b, i
    raise SystemExit(f"extract.py: unterminated definition at {sig_prefix!r}")


def definition_range(lines, needle):
    """[b, e] of the top-level definition whose declarator contains `needle` (a qualified name such as
    `AccountMetaContainer::CheckTres_(`): the return type may sit on the line(s) above the name."""
    hits = [i for i, ln in enumerate(lines) if needle in ln]
    if len(hits) != 1:
        raise SystemExit(f"extract.py: {len(hits)} matches for {needle!r} (expected exactly one definition)")
    b = hits[0]
    while b > 0 and lines[b - 1].strip() and not lines[b - 1].rstrip().endswith((";", "}", "*/")) \
            and not lines[b - 1].lstrip().startswith(("//", "#", "*")):
        b -= 1
    depth, seen = 0, False
    for i in range(hits[0], len(lines)):
        code = re.sub(r'"(\\.|[^"\\])*"', '""', lines[i]).split("//", 1)[0]
        depth += code.count("{") - code.count("}")
        if "{" in code:
            seen = True
        if seen and depth == 0:
            return b, i
    raise SystemExit(f"extract.py: unterminated definition at {needle!r}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit("extract.py: /root/reference is not present (the GPU box uses the prebuilt oracle/_ref/*.so)")
    os.makedirs(args.out, exist_ok=True)
    manifest = []

    def emit(name, rel, ranges, lines, prologue="", epilogue=""):
        body = []
        for b, e in ranges:
            body.append(f"// ---- {rel}:{b + 1}-{e + 1} (verbatim, generated at build time; never committed) ----")
            body.extend(lines[b:e + 1])
        text = prologue + "\n".join(body) + "\n" + epilogue
        with open(os.path.join(args.out, name), "w", encoding="utf-8") as f:
            f.write(text)
        for b, e in ranges:
            manifest.append(f"{name}: {rel}:{b + 1}-{e + 1}")

    # ---- PublicHeader.h: the resource classes --------------------------------------------
    h = read(PH_H)
    b = find_line(h, "/* ----------- Public definitions for all components */")
    e = find_line(h, "[[nodiscard]] constexpr bool IsFinishedStepStatus(", b) - 1
    emit("ph_types.inc", PH_H, [(b, e)], h)

    # ---- PublicHeader.cpp: every definition that does not mention protobuf types ------------
    c = read(PH_CPP)
    first = find_line(c, '#include "crane/PublicHeader.h"') + 1
    keep, dropped = [], []
    for (s, t, chunk) in top_level_defs(c, first):
        text = "\n".join(chunk)
        if "crane::grpc::" in text:
            dropped.append((s, t))
            continue
        if keep and keep[-1][1] + 1 == s:
            keep[-1] = (keep[-1][0], t)
        else:
            keep.append((s, t))
    emit("ph_impl.inc", PH_CPP, keep, c)
    manifest.append("ph_impl.inc: dropped (protobuf conversions): " + ", ".join(f"{s + 1}-{t + 1}" for s, t in dropped))

    # ---- JobScheduler.h: namespace Ctld … end of class SchedulerAlgo ---------------------
    jh = read(JS_H)
    b = find_line(jh, "namespace Ctld {", exact=True)
    e = find_line(jh, "class JobScheduler {", b, exact=True) - 1
    while jh[e].strip() == "":
        e -= 1
    emit("js_types.inc", JS_H, [(b, e)], jh, epilogue="}  // namespace Ctld\n")

    # ---- JobScheduler.cpp: the LocalScheduler functions, NodeSelect, MultiFactorPriority ----
    jc = read(JS_CPP)
    names = [
        "bool SchedulerAlgo::LocalScheduler::CalculateRunningNodesAndStartTime_(",
        "bool SchedulerAlgo::LocalScheduler::GetNodesAndTrySchedule_(",
        "bool SchedulerAlgo::LocalScheduler::Backfill_(",
        "bool SchedulerAlgo::LocalScheduler::TryPreempt_(",
        "void SchedulerAlgo::NodeSelect(",
        "void MultiFactorPriority::GetOrderedJobPtrVec(",
        "void MultiFactorPriority::CalculateFactorBound_(",
        "double MultiFactorPriority::CalculatePriority_(",
    ]
    ranges = [function_range(jc, n) for n in names]
    emit("js_impl.inc", JS_CPP, ranges, jc, prologue="namespace Ctld {\n", epilogue="}  // namespace Ctld\n")

    # ---- AccountMetaContainer.h: MetaResource … end of class AccountMetaContainer ---------------
    ah = read(AMC_H)
    b = find_line(ah, "struct MetaResource {", exact=True)
    e = find_line(ah, "}  // namespace Ctld", b) - 1
    while ah[e].strip() == "":
        e -= 1
    emit("amc_types.inc", AMC_H, [(b, e)], ah, prologue="namespace Ctld {\n", epilogue="}  // namespace Ctld\n")

    # ---- AccountMetaContainer.cpp: the run-limit admission ----------------------------------------
    ac = read(AMC_CPP)
    names = [
        "MetaResource& MetaResource::operator+=(",
        "AccountMetaContainer::CheckAndMallocMetaResource(",
        "AccountMetaContainer::CheckTres_(",
        "AccountMetaContainer::IsUnlimitedTres_(",
        "AccountMetaContainer::CheckQosRunLimitsForEntity_(",
        "AccountMetaContainer::CheckPartitionRunLimitsForEntity_(",
        "AccountMetaContainer::CheckEntityRunLimits_(",
        "AccountMetaContainer::CheckRunLimits_(",
        "AccountMetaContainer::CheckGres_(",
        "AccountMetaContainer::LockAccountStripes_(",
        "AccountMetaContainer::DoMallocResource_(",
    ]
    ranges = sorted(definition_range(ac, n) for n in names)
    emit("amc_impl.inc", AMC_CPP, ranges, ac, prologue="namespace Ctld {\n", epilogue="}  // namespace Ctld\n")

    # ---- CtldPublicDefs.cpp: JobInCtld::SchedulePendingSteps ------------------------------------------
    cc = read(CPD_CPP)
    ranges = [definition_range(cc, "JobInCtld::SchedulePendingSteps(")]
    emit("steps_impl.inc", CPD_CPP, ranges, cc, prologue="namespace Ctld {\n", epilogue="}  // namespace Ctld\n")

    # ---- AccountDefs.h: struct License; LicenseManager.cpp: the pre-pass of NodeSelect ---------------------------------
    ad = read(ACD_H)
    b = find_line(ad, "struct License {", exact=True)
    e = find_line(ad, "};", b, exact=True)
    emit("license_types.inc", ACD_H, [(b, e)], ad, prologue="namespace Ctld {\n", epilogue="}  // namespace Ctld\n")
    lm = read(LM_CPP)
    ranges = [definition_range(lm, "LicenseManager::CheckLicenseCountSufficient(")]
    emit("license_impl.inc", LM_CPP, ranges, lm, prologue="namespace Ctld {\n", epilogue="}  // namespace Ctld\n")

    sha = hashlib.sha256()
    for rel in (PH_H, PH_CPP, JS_H, JS_CPP, AMC_H, AMC_CPP, CPD_CPP, ACD_H, LM_CPP):
        with open(os.path.join(REF, rel), "rb") as f:
            sha.update(f.read())
    manifest.append("sha256 of the nine reference files: " + sha.hexdigest())
    with open(os.path.join(args.out, "MANIFEST.txt"), "w") as f:
        f.write("\n".join(manifest) + "\n")
    print("\n".join(manifest))


if __name__ == "__main__":
    sys.exit(main())
