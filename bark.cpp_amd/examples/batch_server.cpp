// batch_server.cpp - an HTTP front end for the MI355X Bark engine that BATCHES.
//
// Same protocol as the reference's example server (examples/server/server.cpp:128-163): POST /bark with a JSON body {"text": "..."}
// answers with a 24 kHz mono 32-bit float WAV (examples/common.cpp:11-25).  The reference serialises its requests on a mutex around
// bark_generate_audio; here every connection is a thread that hands its text to the request collector of bark_mi355x.h
// (bark_hip_batcher_*): whatever is pending travels through the engine as ONE lock-step batch.  A request may carry "seed": n (default:
// a counter starting at the server's --seed); its audio is what a fresh context loaded with that seed generates, whatever batch it joined.
// Plain POSIX sockets, one thread per connection, Connection: close; no third-party code.
//
//   bark_batch_server -m model.bin [-a 127.0.0.1] [-p 1337] [-s seed] [--max-batch 32] [--max-wait-ms 5] [--streams 1] [--devices 0,1,...] [--temp t] [--fine-temp t]
#include "bark.h"
#include "bark_mi355x.h"
#include "http_util.h"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

using barkhttp::json_string; using barkhttp::json_uint; using barkhttp::wav_f32;

struct Options {
    std::string model, host = "127.0.0.1";
    int port = 1337, max_batch = 32, max_wait_ms = 5, streams = 1;
    std::vector<int> devices;                      // --devices 0,1,...: one context (own copy of the weights) and one worker per listed GPU, one queue
    uint32_t seed = 0;
    float temp = -1.0f, fine_temp = -1.0f;
};

bool send_all(int fd, const char * p, size_t n) {
    while (n) {
        const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k <= 0) return false;
        p += k; n -= (size_t) k;
    }
    return true;
}

void respond(int fd, int status, const char * reason, const char * type, const std::string & body) {
    char head[256];
    const int n = snprintf(head, sizeof(head), "HTTP/1.1 %d %s\r\nContent-Type: %s\r\nContent-Length: %zu\r\nConnection: close\r\n\r\n", status, reason, type, body.size());
    if (send_all(fd, head, (size_t) n)) send_all(fd, body.data(), body.size());
}

std::atomic<uint32_t> next_seed{0};
std::atomic<int> open_connections{0};
constexpr int kMaxConnections = 512;                     // beyond that a connection is answered 503 at once

void serve(int fd, bark_hip_batcher * batcher, int sample_rate) {
    struct Guard { ~Guard() { open_connections.fetch_sub(1); } } guard;
    timeval tv{10, 0};                                       // a client that stops sending does not park this thread for ever
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    std::string req;
    char buf[4096];
    size_t head_end = std::string::npos;
    while (head_end == std::string::npos && req.size() < (1u << 20)) {
        const ssize_t k = ::recv(fd, buf, sizeof(buf), 0);
        if (k <= 0) { ::close(fd); return; }
        req.append(buf, (size_t) k);
        head_end = req.find("\r\n\r\n");
    }
    if (head_end == std::string::npos) { respond(fd, 400, "Bad Request", "text/plain", "bad request"); ::close(fd); return; }
    const std::string head = req.substr(0, head_end);
    size_t content_length = 0;
    {
        std::string lower = head;
        for (char & ch : lower) ch = (char) tolower((unsigned char) ch);
        const size_t p = lower.find("content-length:");
        if (p != std::string::npos) content_length = (size_t) strtoul(head.c_str() + p + 15, nullptr, 10);
    }
    if (content_length > (1u << 20)) { respond(fd, 413, "Payload Too Large", "text/plain", "too large"); ::close(fd); return; }
    std::string body = req.substr(head_end + 4);
    while (body.size() < content_length) {
        const ssize_t k = ::recv(fd, buf, sizeof(buf), 0);
        if (k <= 0) break;
        body.append(buf, (size_t) k);
    }
    if (head.compare(0, 4, "GET ") == 0) {
        respond(fd, 200, "OK", "text/html", "<html>bark batch server: POST /bark {\"text\": \"...\"}</html>");
    } else if (head.compare(0, 11, "POST /bark ") == 0 || head.compare(0, 11, "POST /bark?") == 0) {
        std::string text;
        if (!json_string(body, "text", text)) {
            respond(fd, 400, "Bad Request", "text/plain", "expected a JSON body with a \"text\" string");
        } else {
            uint32_t seed = 0;
            if (!json_uint(body, "seed", seed)) seed = next_seed.fetch_add(1);
            const int64_t ticket = bark_hip_batcher_submit(batcher, text.c_str(), seed);
            int n = ticket > 0 ? bark_hip_batcher_wait(batcher, ticket, nullptr, 0) : -1;     // probe: -(2 + samples)
            if (n <= -2) {
                std::vector<float> pcm((size_t) (-n - 2));
                n = bark_hip_batcher_wait(batcher, ticket, pcm.data(), (int) pcm.size());
                if (n >= 0) respond(fd, 200, "OK", "audio/wav", wav_f32(pcm.data(), n, sample_rate));
            }
            if (n < 0) respond(fd, 500, "Internal Server Error", "text/plain", "Internal Server Error");
        }
    } else {
        respond(fd, 404, "Not Found", "text/plain", "not found");
    }
    ::shutdown(fd, SHUT_RDWR);
    ::close(fd);
}

void usage(const char * argv0) {
    fprintf(stderr, "usage: %s -m model.bin [-a host] [-p port] [-s seed] [--max-batch n (<= 256; the context serves up to 64 at a time)] [--max-wait-ms n] [--streams n (1 .. 4 jobs in flight)] [--devices 0,1,... (one context and one worker per GPU, one queue)] [--temp t] [--fine-temp t]\n", argv0);
}

}  // namespace

int main(int argc, char ** argv) {
    Options o;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](const char * what) -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", what); usage(argv[0]); exit(1); } return argv[++i]; };
        if (a == "-m" || a == "--model") o.model = next("-m");
        else if (a == "-a" || a == "--address") o.host = next("-a");
        else if (a == "-p" || a == "--port") o.port = atoi(next("-p"));
        else if (a == "-s" || a == "--seed") o.seed = (uint32_t) strtoul(next("-s"), nullptr, 10);
        else if (a == "--max-batch") o.max_batch = atoi(next("--max-batch"));
        else if (a == "--max-wait-ms") o.max_wait_ms = atoi(next("--max-wait-ms"));
        else if (a == "--streams") o.streams = atoi(next("--streams"));
        else if (a == "--devices") { for (const char * p = next("--devices"); *p;) { char * e = nullptr; o.devices.push_back((int) strtol(p, &e, 10)); if (e == p) { usage(argv[0]); return 1; } p = *e == ',' ? e + 1 : e; } }
        else if (a == "--temp") o.temp = (float) atof(next("--temp"));
        else if (a == "--fine-temp") o.fine_temp = (float) atof(next("--fine-temp"));
        else { usage(argv[0]); return a == "-h" || a == "--help" ? 0 : 1; }
    }
    if (o.model.empty()) { usage(argv[0]); return 1; }
    signal(SIGPIPE, SIG_IGN);
    bark_context_params params = bark_context_default_params();
    if (o.temp >= 0.0f) params.temp = o.temp;
    if (o.fine_temp >= 0.0f) params.fine_temp = o.fine_temp;
    // one GPU: the context of bark_load_model (+ --streams job streams on clones of it); --devices: a context per listed GPU behind one queue
    std::vector<bark_context *> ctxs;
    if (o.devices.empty()) ctxs.push_back(bark_load_model(o.model.c_str(), params, o.seed));
    else for (int d : o.devices) ctxs.push_back(bark_hip_load_model_on_device(o.model.c_str(), params, o.seed, d));
    for (bark_context * c : ctxs) if (!c) { fprintf(stderr, "%s: could not load the model\n", argv[0]); for (bark_context * x : ctxs) if (x) bark_free(x); return 1; }
    bark_context * ctx = ctxs[0];
    bark_hip_batcher * batcher = ctxs.size() > 1 ? bark_hip_batcher_create_multi(ctxs.data(), (int) ctxs.size(), o.max_batch, o.max_wait_ms)
                                                 : bark_hip_batcher_create_ex(ctx, o.max_batch, o.max_wait_ms, o.streams);
    if (!batcher) { fprintf(stderr, "%s: could not create the request collector\n", argv[0]); for (bark_context * x : ctxs) bark_free(x); return 1; }
    next_seed = o.seed;

    const int lfd = ::socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr{};
    addr.sin_family = AF_INET; addr.sin_port = htons((uint16_t) o.port);
    if (inet_pton(AF_INET, o.host.c_str(), &addr.sin_addr) != 1 || bind(lfd, reinterpret_cast<sockaddr *>(&addr), sizeof(addr)) != 0 || listen(lfd, 128) != 0) {
        fprintf(stderr, "couldn't bind to server socket: hostname=%s port=%d\n", o.host.c_str(), o.port);
        bark_hip_batcher_free(batcher); for (bark_context * x : ctxs) bark_free(x);
        return 1;
    }
    printf("\nbark batch server listening at http://%s:%d (lock-step batches of up to %d requests, %d ms to fill)\n\n", o.host.c_str(), o.port, o.max_batch, o.max_wait_ms);
    fflush(stdout);
    while (true) {
        const int fd = ::accept(lfd, nullptr, nullptr);
        if (fd < 0) continue;
        if (open_connections.fetch_add(1) >= kMaxConnections) {
            open_connections.fetch_sub(1);
            respond(fd, 503, "Service Unavailable", "text/plain", "too many connections");
            ::close(fd);
            continue;
        }
        std::thread(serve, fd, batcher, params.sample_rate).detach();
    }
}
