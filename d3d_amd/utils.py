class Dict(dict):
    """dict with attribute access.  The reference wraps its results in addict.Dict
    (d3d/voxel/__init__.py:1,93-97); callers use both `ret.voxels` and `'aggregates' in ret`."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value
