"""GPU: the LSTM kernels (one direction of one layer; composed into bidirectional multi-layer nn.LSTM semantics by
_unit.lstm_forward) against torch.nn.LSTM on the CPU with the same parameters: outputs 2e-6, input and parameter gradients
2e-5 relative (fp32, different summation order), for the reference's shapes (CnnLSTM: 32 steps of 21 features, H=64,
bidirectional; MLSTM_FCN: 4 layers, H=128) and ragged ones."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src.models._unit import lstm_forward


@pytest.mark.parametrize("S,B,I,H,layers,bidir", [(32, 8, 21, 64, 1, True), (21, 5, 14, 128, 4, True), (7, 3, 5, 9, 2, False),
                                                   (1, 2, 3, 4, 1, True), (21, 64, 12, 128, 2, True)])
def test_lstm_matches_torch(S, B, I, H, layers, bidir):
    torch.manual_seed(S * 7 + H)
    ref = torch.nn.LSTM(I, H, num_layers=layers, bidirectional=bidir, batch_first=False)
    mine = torch.nn.LSTM(I, H, num_layers=layers, bidirectional=bidir, batch_first=False)
    mine.load_state_dict(ref.state_dict())
    mine.cuda()
    x = torch.randn(S, B, I)
    xr = x.clone().requires_grad_(True)
    out, _ = ref(xr)
    dout = torch.randn(out.shape)
    out.backward(dout)
    xg = x.cuda().requires_grad_(True)
    og = lstm_forward(xg, mine)
    og.backward(dout.cuda())
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / max(1e-12, float(b.double().abs().max())))
    assert rel(og.detach().cpu(), out.detach()) < 2e-6
    assert rel(xg.grad.cpu(), xr.grad) < 2e-5
    # >= 512 (t, b) rows: dW_hh comes from the split-precision (3 x bf16) weight-gradient GEMM, fp32-level but not fp32-exact
    ptol = 1e-4 if S * B >= 512 else 2e-5
    for (k, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        assert rel(q.grad.cpu(), p.grad) < ptol, k
