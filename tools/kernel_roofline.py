def dominant_kernel_roofline(device, B):
    raise NotImplementedError("filled in after the first profile")
