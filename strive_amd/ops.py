"""placeholder -- filled in below"""
def mlp_forward(*a, **k): raise NotImplementedError
def gnn_forward(*a, **k): raise NotImplementedError
