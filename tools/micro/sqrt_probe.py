import numpy as np, torch
r = np.float32(5.235566646888401e-08)
t = torch.tensor(r)
print('numpy sqrt', repr(float(np.sqrt(r))), 'torch sqrt', repr(float(torch.sqrt(t))), 'torch f64->f32', repr(float(np.float32(np.sqrt(np.float64(r))))))
tr = torch.tensor(r, requires_grad=True)
print('torch sqrt (grad)', repr(float(torch.sqrt(tr))))
er = torch.sqrt(t).to(torch.float64); expo = torch.tensor(1 / 5).to(torch.float64)
print('pow torch', repr(float(er ** expo)), 'pow libm', repr(float(er) ** float(expo)))
print(torch.__config__.show()[:600])
