"""Host-side logic of deeprl_amd/nets.py that needs no GPU: the ReLU-mask hand-over between layer Functions and the in-place
parameter-gradient switch (round 4)."""
import torch


def test_relu_mask_mark_is_consumed_once_and_checks_address_and_size():
    from deeprl_amd import nets
    dx = torch.zeros(6, 4)
    nets._mark_masked(dx)
    assert nets._already_masked(dx.view(6, 4)) is True          # the layer below receives that very tensor (a view of it)
    assert nets._already_masked(dx) is False                    # ... once
    nets._mark_masked(dx)
    other = torch.zeros(6, 4)
    assert nets._already_masked(other) is False                 # another tensor: masked as before
    assert nets._already_masked(dx) is False                    # and the stale mark is gone
    nets._mark_masked(dx)
    assert nets._already_masked(dx[:3]) is False                # same address, different extent: not the marked gradient


def test_direct_param_grads_switch_nests_and_restores():
    from deeprl_amd import nets
    assert nets._DIRECT[0] is False
    with nets.direct_param_grads():
        assert nets._DIRECT[0] is True
        with nets.direct_param_grads(False):
            assert nets._DIRECT[0] is False
        assert nets._DIRECT[0] is True
    assert nets._DIRECT[0] is False
    try:
        with nets.direct_param_grads():
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert nets._DIRECT[0] is False


def test_grad_slot_needs_a_matching_dense_device_gradient():
    from deeprl_amd import nets
    p = torch.nn.Parameter(torch.zeros(3, 5))
    assert nets._grad_slot(p) is None and nets._grad_slot(None) is None       # no .grad yet
    p.grad = torch.zeros(3, 5)
    assert nets._grad_slot(p) is None                                         # a CPU gradient is never written in place


def test_nature_conv_body_marks_the_layers_that_read_a_fused_relu():
    from deeprl_amd import nets
    body = nets.NatureConvBody()
    assert not getattr(body.conv1, "input_is_relu", False)
    assert body.conv2.input_is_relu and body.conv3.input_is_relu and body.fc4.input_is_relu
    noisy = nets.NatureConvBody(noisy_linear=True)
    assert not getattr(noisy.fc4, "input_is_relu", False)
