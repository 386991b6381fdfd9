"""Per-call options of the operators (d3d_amd.options) for tests that reach the operators through several layers (autograd
Functions, VoxelGenerator, the reference's own test code): set_opts(...) binds them to the test's calling context,
conftest.py unwinds them when the test ends.  Nothing here touches module state of the product."""
_tokens = []


def set_opts(**kw):
    from d3d_amd import options
    _tokens.append(options.push(**kw))


def cur_opts():
    from d3d_amd import options
    return options.current()


def unwind():
    from d3d_amd import options
    while _tokens:
        options.pop(_tokens.pop())
