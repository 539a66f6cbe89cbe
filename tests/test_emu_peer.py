"""CPU tier: the peer-store exchange kernels (csrc/sc_kernels_peer.h) in host emulation with ONE rank -- window header
layout, the workgroup ticket, the epoch advancing per call, ragged block sizes, argument checks.  (Several ranks need
several processes on a device: tests/test_gpu_peer_exchange.py.)"""
import pytest
import torch

from engine_runner import emu_lib


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.mark.parametrize("n", [4, 1028, 40000])
def test_one_rank_exchange_is_a_copy_and_epochs_advance(lib, n):
    ptr, handle = lib.peer_window_alloc(n * 4)
    assert len(handle) == 64 and lib.peer_window_open(handle) == ptr           # emulation: the handle is the pointer
    send = torch.randn(1, n)
    for rep in range(3):
        recv = torch.full_like(send, float("nan"))
        lib.peer_all_to_all(1, 0, n * 4, [ptr], send.data_ptr(), recv.data_ptr())
        assert torch.equal(recv, send)
        send = send + 1.0
    import ctypes
    flags = (ctypes.c_uint64 * 8).from_address(ptr)
    epoch = ctypes.c_uint64.from_address(ptr + 256).value
    ticket = ctypes.c_uint32.from_address(ptr + 264).value
    assert epoch == 3 and flags[0] == 3 and ticket == 0
    lib.peer_window_free(ptr)


def test_argument_checks(lib):
    ptr, _ = lib.peer_window_alloc(1024)
    a, b = torch.zeros(1, 64), torch.zeros(1, 64)
    with pytest.raises(RuntimeError):
        lib.peer_all_to_all(1, 0, 24, [ptr], a.data_ptr(), b.data_ptr())          # not a multiple of 16
    with pytest.raises(RuntimeError):
        lib.peer_all_to_all(9, 0, 16, [ptr], a.data_ptr(), b.data_ptr())          # more than one node's 8 ranks
    with pytest.raises(RuntimeError):
        lib.peer_all_to_all(2, 0, 16, [ptr], a.data_ptr(), b.data_ptr())          # peer 1's window is not mapped
    lib.peer_window_free(ptr)


def test_a_wait_under_a_spin_budget_ends_with_an_error_word_instead_of_hanging(lib):
    """k_peer_wait spun without bound, so a dead peer (or a window that is not coherent) turned the set-up
    self-test into a synchronize that never returns.  With a budget on the window (sc_peer_window_control) the wait gives
    up and leaves 1 + the index of the peer whose flag never came; reading the word clears it.  Here: two "ranks" in one
    emulated process, only rank 0 ever calls the exchange."""
    w0, _ = lib.peer_window_alloc(1024)
    w1, _ = lib.peer_window_alloc(1024)
    assert lib.peer_window_control(w0) == 0                                         # fresh window: no error, budget unbounded
    lib.peer_window_control(w0, 1)                                                  # 1 ms of the (emulated) clock
    send, recv = torch.ones(2, 8), torch.zeros(2, 8)
    lib.peer_all_to_all(2, 0, 32, [w0, w1], send.data_ptr(), recv.data_ptr())       # returns: the wait timed out on peer 1
    assert lib.peer_window_control(w0) == 2                                         # 1 + peer 1
    assert lib.peer_window_control(w0) == 0                                         # cleared by the read
    assert torch.equal(recv[0], send[0])                                            # its own block did arrive
    lib.peer_window_free(w0)
    lib.peer_window_free(w1)
