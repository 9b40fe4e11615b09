"""SYNTAX ONLY, NOT VERIFICATION (tests/adapter_syntax/README.md): adapter/*.cpp parsed and type-checked by g++ against this
repository's own minimal mock of the OpenVINO declarations it touches -- the image has no OpenVINO developer package, so the
real build of the adapter cannot run here.  Catches typos and signature drift against include/ovtk_amd.h; says nothing about
behaviour (that is what the C-ABI parity tests are for)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_adapter_sources_parse_against_the_mock_headers():
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", str(ROOT / "tests" / "adapter_syntax" / "mock"),
                        "-I", str(ROOT / "include"), str(ROOT / "adapter" / "ops.cpp"), str(ROOT / "adapter" / "extension.cpp"),
                        str(ROOT / "adapter" / "fuse_pass.cpp")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]


def test_extension_list_and_factory_cover_the_same_ops():
    """Every op class of adapter/ops.hpp is in the OPENVINO_CREATE_EXTENSIONS list and in create_tokenizer_node's chain -- the
    reference's two entry points (src/ov_extension.cpp:72-109, src/tokenizers_factory.cpp:23-74) for the ops this library replaces."""
    import re
    hpp = (ROOT / "adapter" / "ops.hpp").read_text()
    ext = (ROOT / "adapter" / "extension.cpp").read_text()
    ops = set(re.findall(r'OPENVINO_OP\("(\w+)"', hpp)) | set(re.findall(r"OVTK_ADAPTER_STATELESS_OP\((\w+)\);", hpp))
    ops.discard("Name")
    fused = {o for o in ops if o.startswith("OvtkFused")}   # the nodes adapter/fuse_pass.cpp creates: never read from an IR, not in the lists
    assert fused == {"OvtkFusedSplitBPE", "OvtkFusedSplitWordpiece", "OvtkFusedDetokenize"}
    ops -= fused
    assert len(ops) == 15 and "StringTensorPack" in ops
    listed = set(re.findall(r"ov::OpExtension<(\w+)>", ext))
    made = set(re.findall(r'op_type == "(\w+)"', ext))
    assert listed == ops and made == ops


def test_fuse_pass_covers_the_chains_of_the_python_recogniser():
    """adapter/fuse_pass.cpp and openvino_tokenizers_amd/pipeline.py fuse() name the same chains (the Python one is what runs in the tests:
    tests/test_pipeline_fuse.py)."""
    cpp = (ROOT / "adapter" / "fuse_pass.cpp").read_text()
    py = (ROOT / "openvino_tokenizers_amd" / "pipeline.py").read_text()
    for entry in ("ovtk_encode_run", "ovtk_wordpiece_encode_run", "ovtk_detokenize_run"):
        assert entry in py and entry in (ROOT / "adapter" / "ops.cpp").read_text()
    for node in ("FusedSplitBPE", "FusedSplitWordpiece", "FusedDetokenize"):
        assert node in cpp
