"""smoke(): one tiny forward + backward + Adam step of the DualQ-SELD-TCN on cuda:0, checked against
the CPU oracle (oracle/ is test infrastructure: it is imported here only as the checker)."""
import torch


def smoke():
    from . import model as M
    from . import train as T
    from oracle import seld_oracle as O

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a HIP device")
    dev = torch.device("cuda:0")
    kw = dict(time_dim=64, freq_dim=128, input_channels=8, output_classes=14, domain='DQ', domain_classifier='DQ',
              cnn_filters=[16, 16, 16], pool_size=[[8, 2], [8, 2], [2, 2]], pool_time='TCN', D=[10],
              dilation_mode='fibonacci', G=32, U=16, V=[16, 16], V_kernel_size=3, fc_layers=[16],
              fc_activations='linear', fc_dropout='Last', dropout_perc=0.0, spatial_dropout_rate=0.0,
              class_overlaps=3, use_bias_conv=0, use_bias_linear=1, batch_norm='BN')
    torch.manual_seed(1)
    m = M.SELD_Model(**kw)
    O.closed_form_fill_(list(m.state_dict().items()))
    sd64 = {k: v.detach().double().clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    x = O.closed_form_input((2, 8, 128, 64))
    target = torch.cat(((torch.sin(0.7 * torch.arange(2 * 8 * 42.)) > 0.8).float().view(2, 8, 42),
                        0.9 * torch.sin(0.013 * torch.arange(2 * 8 * 126.)).view(2, 8, 126)), 2)
    opt = T.FlatAdam(m.parameters(), lr=1e-4)
    opt.zero_grad()
    sed, doa = m(x.to(dev))
    loss = T.seld_loss_fn(sed, doa, target.to(dev), 42, 1.0, 5.0)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()

    cfg = O.SeldConfig(**{k: v for k, v in kw.items()})
    for v in sd64.values():
        v.requires_grad_(v.is_floating_point())
    sed_r, doa_r = O.seld_forward(sd64, cfg, x.double(), train=True, mode="explicit")
    loss_r = O.seld_loss(sed_r, doa_r, target.double(), 42)
    loss_r.backward()
    err = max((sed.detach().cpu().double() - sed_r.detach()).abs().max().item(),
              (doa.detach().cpu().double() - doa_r.detach()).abs().max().item())
    assert err < 1e-3, f"smoke: output mismatch {err}"
    assert abs(loss.item() - loss_r.item()) < 1e-4 * max(1.0, abs(loss_r.item())), (loss.item(), loss_r.item())
    g = dict(m.named_parameters())["seld_block.cnn.0.0.r_weight"].grad.cpu().double()
    gr = sd64["seld_block.cnn.0.0.r_weight"].grad
    assert (g - gr).abs().max().item() <= 1e-3 * max(gr.abs().max().item(), 1e-6), "smoke: gradient mismatch"
    print(f"smoke ok: max|out - oracle| = {err:.2e}, loss = {loss.item():.6f} (oracle {loss_r.item():.6f})")
