"""The checker the GPU tests of test_gpu_launch_order.py rely on (harness.launch_order_violations), on hand-written logs."""
import harness as h


def test_the_checker_finds_an_unordered_pair():
    ok = "L 0 a 0x10:w\nR 0 0xe1\nW 1 0xe1\nL 1 b 0x10:r\nL 1 c 0x20:w\nR 1 0xe2\nW 0 0xe2\nL 0 d 0x20:r 0x10:w\n"
    assert h.launch_order_violations(ok) == []
    assert len(h.launch_order_violations(ok.replace("W 1 0xe1\n", ""))) == 1          # b reads 0x10 beside a's write
    assert len(h.launch_order_violations(ok.replace("W 0 0xe2\n", ""))) == 2          # d against c's write and b's read
    assert h.launch_order_violations("L 0 a 0x10:w~\nL 1 b 0x10:r~\n") == []          # relaxed counts
    assert len(h.launch_order_violations("L 1 a 0x10:w\nH 0\n")) == 1                 # host operation beside a launch
    assert len(h.launch_order_violations("H 0\nL 1 a 0x10:w\n")) == 1
    assert h.launch_order_violations("H 0\nR 0 0xe\nW 1 0xe\nL 1 a 0x10:w\nR 1 0xf\nW 0 0xf\nH 0\n") == []
    assert h.launch_order_violations("L 5 transfer 0x10:w:0-8\nY 0\nH 0\nL 1 a 0x10:r\n".replace("L 1 a", "R 0 0xe\nW 1 0xe\nL 1 a")) == []
    # rows: a transfer into rows 0-8 beside a kernel on rows 8-40 is fine, beside one that reads rows 5-43 it is not
    assert h.launch_order_violations("L 5 transfer 0x10:w:0-8\nL 0 spatial 0x10:r:8-40\n") == []
    assert len(h.launch_order_violations("L 5 transfer 0x10:w:0-8\nL 0 spatial 0x10:r:5-43\n")) == 1
