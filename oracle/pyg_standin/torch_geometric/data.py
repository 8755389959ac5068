class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return self.x.size(0)

    def to(self, device):
        for k, v in list(vars(self).items()):
            if hasattr(v, "to"):
                setattr(self, k, v.to(device))
        return self
