"""The drop-in's cached parameter list (ddif/models/sr3_dwt.py `_param_cache`): the training loop asks for `named_parameters()` several times per iteration and
the walk over 702 parameters costs milliseconds of Python with the GPU idle behind it, so it is walked once -- and must notice every way the set can change."""
import torch

import golden_cases as gc
from ddif_testlib import CTOR_KEYS


def _net():
    from ddif.models.sr3_dwt import UNetSR3

    cfg = gc.cfg_for("wv3")
    return UNetSR3(**{k: cfg[k] for k in CTOR_KEYS})


def test_cached_list_is_the_walked_list_and_is_reused():
    net = _net()
    a = net.named_parameter_list()
    assert [(n, id(p)) for n, p in a] == [(n, id(p)) for n, p in net.named_parameters()]
    assert net.named_parameter_list() is a  # no second walk
    assert net._signature() == tuple((p.data_ptr(), p._version) for p in net.parameters())


def test_registration_anywhere_and_apply_invalidate_the_cache():
    net = _net()
    a = net.named_parameter_list()
    other = torch.nn.Linear(2, 2)  # a registration in an unrelated module bumps the process-wide epoch: the list is re-walked, equal, a new object
    b = net.named_parameter_list()
    assert b is not a and [n for n, _ in b] == [n for n, _ in a]
    del other
    net.register_parameter("extra_test_parameter", torch.nn.Parameter(torch.zeros(3)))
    c = net.named_parameter_list()
    assert len(c) == len(a) + 1 and "extra_test_parameter" in [n for n, _ in c]
    del net._parameters["extra_test_parameter"]  # (behind torch's back: only `_apply` / a registration re-walks)
    net = net.to(torch.device("cpu"))  # `_apply`
    d = net.named_parameter_list()
    assert [n for n, _ in d] == [n for n, _ in a]


def test_signature_sees_in_place_updates_and_moved_storage():
    net = _net()
    s0 = net._signature()
    p = net.named_parameter_list()[5][1]
    with torch.no_grad():
        p.add_(1.0)
    s1 = net._signature()
    assert s1 != s0
    p.data = p.data.clone()  # storage moved, same Parameter object
    assert net._signature() != s1
