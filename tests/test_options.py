"""d3d_amd.options: per-call options live in the calling context -- never in module state."""
import threading

import pytest


def test_scope_nests_and_unwinds():
    from d3d_amd import options
    assert options.current().voxel_flags == 0 and options.current().poison is False
    with options.scope(voxel_flags=1, poison=True):
        assert options.current().voxel_flags == 1 and options.current().poison is True
        with options.scope(nms_flags=8):
            cur = options.current()
            assert (cur.voxel_flags, cur.nms_flags, cur.poison) == (1, 8, True)
        assert options.current().nms_flags == 0
    assert options.current().voxel_flags == 0 and options.current().poison is False


def test_unknown_option_is_a_type_error():
    from d3d_amd import options
    with pytest.raises(TypeError):
        options.push(no_such_option=1)
    assert options.current().voxel_flags == 0


def test_threads_do_not_see_each_others_options():
    """two threads bind different flags at the same time; each reads its own, the main thread keeps the defaults"""
    from d3d_amd import options
    seen, gate = {}, threading.Barrier(2)

    def run(name, flags):
        with options.scope(voxel_flags=flags, nms_flags=flags << 1):
            gate.wait()                                  # both are inside their scopes now
            cur = options.current()
            seen[name] = (cur.voxel_flags, cur.nms_flags)
            gate.wait()
    ts = [threading.Thread(target=run, args=("a", 1)), threading.Thread(target=run, args=("b", 4))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert seen == {"a": (1, 2), "b": (4, 8)}
    assert options.current().voxel_flags == 0 and options.current().nms_flags == 0


def test_operators_have_no_option_globals():
    """the module-level switches of rounds 1-3 are gone (a second thread could flip them under a running call)"""
    from d3d_amd import box, voxel
    for mod, names in ((voxel, ("default_flags", "poison_outputs")), (box, ("default_nms_flags", "default_iou_flags", "poison_outputs"))):
        for n in names:
            assert not hasattr(mod, n), "%s.%s" % (mod.__name__, n)
