"""The scan kernel's launch shapes against the compiler's own figures, without a GPU.

knox_kernels.h `scan_shape` is the ONE table both sides of a scan launch read: the kernel parks deferred bitset words in LDS
sized for `per_cu` workgroups per CU, run_fused launches `per_cu x CUs` workgroups, and the chip-wide write phases of the
deferral only line up when that whole grid is resident.  Whether it is, is decided by registers and LDS --
which the build records per instantiation (`-Rpass-analysis=kernel-resource-usage` -> knoxdb_amd/csrc/knox_scan_w<W>.res:
VGPRs, LDS bytes, "Occupancy [waves/SIMD]" = workgroups of 4 waves per CU).  Round 4 found six instantiations whose table
entry exceeded what fits (two-leaf scans of 1-byte columns at 256 VGPRs: one workgroup per CU where three were launched);
this test keeps table and build together.  On the GPU the same question is asked of the runtime at first launch
(knox_scan_occupancy_mismatches, tests/test_gpu_defer.py)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from knoxdb_amd import _lib

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "knoxdb_amd", "csrc")
A_ISUM, A_FSUM, FL_FLOAT = 1, 2, 2


def instantiations(width):
    path = os.path.join(CSRC, "knox_scan_w%d.res" % width)
    assert os.path.exists(path), "build first (__graft_entry__.build): the scan objects leave their resource figures beside them"
    blocks = open(path).read().split("Function Name: ")[1:]
    names = [b.split()[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    out = []
    for b, d in zip(blocks, dem):
        m = re.search(r"scan_kernel<(\d+), (\d+), (\d+)u, (true|false), (true|false), (\d+)u>", d)
        if not m:
            continue
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        out.append({"W": int(m.group(1)), "NL": int(m.group(2)), "AGG": int(m.group(3)), "SEP": m.group(4) == "true", "ANYF": m.group(5) == "true",
                    "FLV": int(m.group(6)), "vgprs": g("VGPRs"), "lds": g(r"LDS Size \[bytes/block\]"), "occ": g(r"Occupancy \[waves/SIMD\]"),
                    "scratch": g(r"ScratchSize \[bytes/lane\]"), "name": d.strip()})
    return out


@pytest.mark.parametrize("width", [8, 4, 2, 1])
def test_every_deferring_instantiation_fits_its_launch_shape(width):
    L = _lib.lib()
    inst = instantiations(width)
    assert len(inst) >= 30 and all(i["W"] == width for i in inst)
    checked = 0
    for i in inst:
        per, park = C.c_int(), C.c_uint32()
        heavy = (i["AGG"] & ~A_ISUM) != 0
        assert L.knox_scan_launch_shape(width, i["NL"], int(heavy), int(i["FLV"] == FL_FLOAT), int((i["AGG"] & A_FSUM) != 0), C.byref(per), C.byref(park)) == 0
        # the kernel's static LDS is the parked stage (4 waves x park words x 4 B, at least one tile's words) + a few hundred bytes
        if park.value:
            assert i["lds"] >= 16 * park.value, i["name"]
            assert i["lds"] * per.value <= 160 * 1024, (i["name"], i["lds"], per.value)
            # what the launch asks for is resident: registers and LDS allow at least per_cu workgroups per CU
            assert i["occ"] >= per.value, "%s: %d VGPRs, %d B of LDS -> %d workgroups per CU, the launch shape assumes %d" % (
                i["name"], i["vgprs"], i["lds"], i["occ"], per.value)
            checked += 1
    assert checked >= 20


def test_the_headline_kernels_do_not_spill():
    """BASELINE configs[1] / configs[2] / configs[3]-uint64: no scratch, two workgroups per CU"""
    by = {(i["NL"], i["AGG"], i["SEP"], i["ANYF"], i["FLV"]): i for i in instantiations(8)}
    for key in ((1, 1, False, False, 0), (2, 14, False, True, 2), (1, 0, False, False, 0)):
        i = by[key]
        assert i["scratch"] == 0 and i["occ"] >= 2, i


def test_no_scan_instantiation_keeps_anything_in_scratch_memory():
    """Round 4: the three- and four-leaf aggregate-heavy instantiations kept their whole r[NL][4] block of loaded values in
    scratch memory (ScratchSize 208 / 272 bytes per lane, twelve / sixteen 16-byte scratch stores in EVERY wave iteration) because
    the rare exact min / max path indexed the array with a run-time loop variable; they ran at 0.55-0.59 of what they do now
    (tools/ab_kernels.py on identical buffers: 3.6-3.8 -> 6.4-6.7 TB/s).  The build's resource remarks would have shown it:
    hold every instantiation of every width to ScratchSize 0 and no spilled register."""
    n = 0
    for w in (8, 4, 2, 1):
        for i in instantiations(w):
            assert i["scratch"] == 0, i
            n += 1
    assert n >= 250


def test_no_kernel_of_any_family_keeps_anything_in_scratch_memory():
    """the same figures for every other kernel family (knox_agg / window / bitset / bytes / bloom / decode / sort / fold .res, left by the
    build): no kernel spills a register or holds an array in scratch memory"""
    seen = 0
    for fam in ("knox_agg", "knox_window", "knox_bitset", "knox_bytes", "knox_bloom", "knox_decode", "knox_sort", "knox_fold"):
        path = os.path.join(CSRC, fam + ".res")
        assert os.path.exists(path), "build first (__graft_entry__.build): %s.res" % fam
        for b in open(path).read().split("Function Name: ")[1:]:
            g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
            assert g(r"ScratchSize \[bytes/lane\]") == 0 and g("VGPRs Spill") == 0, (fam, b.split()[0])   # (SGPRs spill into VGPR lanes, not memory)
            seen += 1
    assert seen >= 250
