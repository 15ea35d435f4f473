"""The RCCL entry points the engine binds with dlopen (csrc/cs_rccl.hip.inc) are declared there by hand: the
library is optional at build time.  This checks those hand-written declarations against the installed rccl.h:
argument lists, the by-value 128-byte unique id, and the enum values the engine passes."""
import os
import subprocess
import sys
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RCCL_H = "/opt/rocm/include/rccl/rccl.h"


@pytest.mark.skipif(not os.path.exists(RCCL_H), reason="no rccl.h on this machine")
def test_hand_written_rccl_declarations_match_the_header(tmp_path):
    src = tmp_path / "rccl_abi.cpp"
    src.write_text(textwrap.dedent("""
        #include <type_traits>
        #include <cstddef>
        #include <rccl/rccl.h>
        #include "crowdstep.h"
        // the typedefs of namespace rccl_api in rmf_crowdsim_amd/csrc/cs_rccl.hip.inc, restated
        struct UniqueId { char internal[CS_RCCL_UNIQUE_ID_BYTES]; };
        typedef void* Comm;
        static_assert(sizeof(UniqueId) == sizeof(ncclUniqueId) && alignof(UniqueId) == alignof(ncclUniqueId), "unique id");
        static_assert(CS_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id bytes");
        static_assert(std::is_pointer<ncclComm_t>::value && sizeof(ncclComm_t) == sizeof(Comm), "communicator handle");
        static_assert(ncclSuccess == 0 && ncclUint8 == 1 && ncclInt32 == 2 && ncclMax == 2, "enum values");
        static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int) &&
                      sizeof(ncclResult_t) == sizeof(int), "enums are ints");
        static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId");
        static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value,
                      "ncclCommInitRank");
        static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
        static_assert(std::is_same<decltype(&ncclSend),
                                   ncclResult_t (*)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value,
                      "ncclSend");
        static_assert(std::is_same<decltype(&ncclRecv),
                                   ncclResult_t (*)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclRecv");
        static_assert(std::is_same<decltype(&ncclAllReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t,
                                                                             ncclRedOp_t, ncclComm_t, hipStream_t)>::value,
                      "ncclAllReduce");
        static_assert(std::is_same<decltype(&ncclAllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t,
                                                                             ncclComm_t, hipStream_t)>::value,
                      "ncclAllGather");
        static_assert(std::is_same<decltype(&ncclGroupStart), ncclResult_t (*)()>::value &&
                      std::is_same<decltype(&ncclGroupEnd), ncclResult_t (*)()>::value, "group calls");
        static_assert(std::is_same<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>::value, "error string");
        int main() { return 0; }
    """))
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        "-I" + os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_the_engine_declares_what_the_check_restates():
    """(the check above restates the typedefs: keep it honest against the source)"""
    text = open(os.path.join(ROOT, "rmf_crowdsim_amd", "csrc", "cs_rccl.hip.inc")).read()
    for needle in ("typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);",
                   "typedef int (*SendFn)(const void*, size_t, int, int, Comm, hipStream_t);",
                   "typedef int (*RecvFn)(void*, size_t, int, int, Comm, hipStream_t);",
                   "typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);",
                   "typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);",
                   "constexpr int kUint8 = 1, kInt32 = 2, kMax = 2;"):
        assert needle in text, needle
