import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    (o1, l1), (o2, l2) = a[k], b[k]
    dl = (l1 - l2).abs()
    print(k, "o maxdiff", float((o1 - o2).abs().max()), "lse maxdiff", float(dl.max()), "at", int(dl.argmax()), "shape", tuple(l1.shape),
          "lse1", l1.flatten()[int(dl.argmax())].item(), "lse2", l2.flatten()[int(dl.argmax())].item())
