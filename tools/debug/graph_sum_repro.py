"""Does a captured torch split reduction (x.sum(0) over thousands of rows) replay correctly?  Stand-alone: torch only."""
import torch

dev = 'cuda:0'
torch.manual_seed(0)
for rows, cols in ((9216, 64), (9216, 192), (20736, 64), (1024, 64), (73728, 8)):
    x = torch.randn(rows, cols, device=dev)
    y = torch.zeros(cols, device=dev)
    pre = torch.randn(rows, cols, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        z = x * 1.0 + pre * 0.0        # a producer kernel in front, as in a backward pass
        y.copy_(z.sum(0))
    bad = []
    for it in range(12):
        x.copy_(torch.randn(rows, cols, device=dev))
        want = x.double().sum(0).float()
        if it % 2:
            torch.cuda.synchronize()
            x[0, :1].cpu()
        g.replay()
        torch.cuda.synchronize()
        err = (y - want).abs().max().item()
        if err > 1e-2:
            bad.append((it, round(err, 4)))
    print(f'sum(0) of [{rows}, {cols}] under replay: ' + ('ok' if not bad else f'WRONG at replays {bad}'))
