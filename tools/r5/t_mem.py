import time, torch
torch.cuda.init(); x = torch.empty(1<<20, device="cuda")
for f, name in ((lambda: torch.cuda.mem_get_info(0), "mem_get_info"), (lambda: torch.cuda.memory_reserved(0), "memory_reserved"), (lambda: torch.cuda.memory_allocated(0), "memory_allocated"), (lambda: torch.empty(1200 << 20, dtype=torch.uint8, device="cuda"), "empty1.2G")):
    f(); t = time.perf_counter()
    for _ in range(200): f()
    print(name, round((time.perf_counter() - t) / 200 * 1e6, 1), "us")
