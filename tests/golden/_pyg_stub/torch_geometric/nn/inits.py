def ones(*a, **k): pass
def zeros(t=None, *a, **k):
    if t is not None: t.data.fill_(0)
def glorot(*a, **k): pass
