"""(The gateway EXECUTES in tests/test_gpu_mex_gateway.py, against the functional shim tests/mex_shim/mex_shim.cpp.)
The MATLAB boundary as files: mex/cp_mex.cpp parses against the prototype mex.h + the real cpengine.h, covers the
single-GPU entry points of the header, and every command the matlab/*.m shims send is one the gateway dispatches."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gateway():
    return open(os.path.join(ROOT, "mex", "cp_mex.cpp")).read()


def test_gateway_syntax():
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", "mex/proto", "-I", "include",
                           "mex/cp_mex.cpp"], cwd=ROOT)


def test_gateway_covers_single_gpu_entry_points():
    hdr = open(os.path.join(ROOT, "include", "cpengine.h")).read()
    syms = set(re.findall(r"\b(cp_[a-z_0-9]+)\s*\(", hdr))
    # multi-GPU staging, stream plumbing, streaming/bench hooks and the allocation-free variant are for C/Python callers
    skip = {s for s in syms if s.startswith(("cp_mg_", "cp_iterate_"))} | {
        "cp_set_stream", "cp_stream_copy_gbps", "cp_singular_values_static", "cp_kernel_time_ms"}
    src = _gateway()
    missing = sorted(s for s in syms - skip if not re.search(r"\b%s\s*\(" % s, src))
    assert not missing, missing


def test_shims_use_only_dispatched_commands():
    src = _gateway()
    handled = set(re.findall(r'strcmp\(cmd, "([a-z_]+)"\)', src))
    used = set()
    for f in glob.glob(os.path.join(ROOT, "matlab", "*.m")):
        used |= set(re.findall(r"cp_mex\('([a-z_]+)'", open(f).read()))
    assert used and used <= handled, sorted(used - handled)


def test_prototype_header_is_labelled():
    head = open(os.path.join(ROOT, "mex", "proto", "mex.h")).read(400)
    assert "PROTOTYPE" in head and "NOT MATLAB" in head


# MATLAB builtins / language functions the shims may call (everything else must be shipped in matlab/ or be cp_mex)
_MATLAB_BUILTINS = {
    "abs", "any", "all", "cat", "cd", "ceil", "cell", "cellfun", "char", "class", "containers", "Map", "cumsum", "double", "end", "error", "exist",
    "feval", "fieldnames", "fileparts", "find", "floor", "fprintf", "func2str", "isa", "ischar", "iscell", "isempty", "isequal", "isfield",
    "isKey", "isnumeric", "isreal", "isscalar", "isstruct", "length", "log2", "max", "mean", "mfilename", "min", "mod", "nargin", "nargout", "norm",
    "numel", "onCleanup", "ones", "permute", "prod", "repmat", "reshape", "round", "size", "sprintf", "sqrt", "squeeze", "str2func",
    "strcmp", "strcmpi", "struct", "sum", "svd", "uint64", "int32", "warning", "which", "zeros", "true", "false", "if", "for", "while", "switch",
    "elseif", "function", "return", "case", "otherwise", "disp", "isnan", "nnz", "sign", "exp", "sin", "cos", "pi", "linspace", "meshgrid",
    "ndgrid", "conv2", "kron", "sparse", "speye", "spdiags", "blkdiag", "randn", "rand", "tic", "toc", "logical", "numel", "validateattributes",
    "inputname", "varargin", "varargout", "lower", "upper", "strrep", "regexp", "ismember", "unique", "sort", "flipud", "fliplr", "vertcat", "horzcat",
    "mat2str", "num2str", "int2str", "datestr", "now", "save", "load", "fullfile", "isfile", "isfolder", "addpath", "genpath", "rmpath", "path"}


def test_every_function_the_shims_call_is_shipped_or_builtin():
    """matlab/chambolle_pock.m called chambolle_pock_matlab, which no file provided. Every identifier used as a
    call in matlab/*.m must be a file of matlab/, cp_mex, a local function of the same file, a variable of that file, or a
    MATLAB builtin; and no *_matlab delegate is left."""
    files = sorted(glob.glob(os.path.join(ROOT, "matlab", "*.m")))
    shipped = {os.path.splitext(os.path.basename(f))[0] for f in files} | {"cp_mex"}
    bad = {}
    for f in files:
        src = open(f).read()
        src_j = re.sub(r"\.\.\.[^\n]*\n", " ", src)                                                     # join continuation lines
        code = "\n".join(re.sub(r"'[^'\n]*'", "''", ln).split("%")[0] for ln in src_j.splitlines() if not ln.lstrip().startswith("%"))   # no strings, no comments
        local = set(re.findall(r"^\s*function\s+(?:[^=\n]*=\s*)?([A-Za-z_]\w*)", code, flags=re.M))
        assigned = set(re.findall(r"([A-Za-z_]\w*)\s*(?:\([^=\n]*\)|\{[^=\n]*\})?\s*=(?!=)", code)) | \
            set(re.findall(r"\[([^\]\n]*)\]\s*=", code) and re.findall(r"[A-Za-z_]\w*", " ".join(re.findall(r"\[([^\]\n]*)\]\s*=", code))))
        params = set(re.findall(r"[A-Za-z_]\w*", " ".join(re.findall(r"function[^\n]*\(([^)]*)\)", code))))
        loopv = set(re.findall(r"\bfor\s+([A-Za-z_]\w*)\s*=", code))
        calls = set(re.findall(r"(?<![\w.@])([A-Za-z_]\w*)\s*\(", code))
        unknown = {c for c in calls - shipped - local - assigned - params - loopv - _MATLAB_BUILTINS if not c.endswith("_matlab") or True}
        unknown = {c for c in unknown if c not in ("c",)}
        if unknown:
            bad[os.path.basename(f)] = sorted(unknown)
        assert "_matlab(" not in src, f
    assert not bad, bad
