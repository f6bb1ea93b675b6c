/*
 * cassiesim.c -- the UDP lock-step simulator server on top of the MI355X library.
 *
 * Plays the role of reference example/cassiesim.c:193-293: a controller sends packed cassie_user_in_t (or, with -x,
 * pd_in_t) datagrams, every received packet advances the simulator by one 0.5 ms step and the packed cassie_out_t (or
 * state_out_t) goes back to the sender; -r runs continuously at 2 kHz on the newest packet instead of in lock step and
 * reports when a step took longer than the time it simulates.  Written against the same C ABI the reference program
 * uses (include/cassiemujoco.h, include/udp.h and the pack / unpack functions of the Agility library); there is no
 * visualiser here (-v is accepted and ignored: rendering is out of scope).
 *
 *   cassiesim [-a address] [-p port] [-m modelfile] [-x] [-r] [-h] [-l log] [-q statelog] [-n steps] [-s statefile]
 *
 *   -n  stop after that many steps (the reference runs until its window closes)
 *   -s  write the final simulator state with cassie_state_save when stopping (and load it at start if it exists)
 */
#include <getopt.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "cassiemujoco.h"
#include "udp.h"

/* wire sizes (float32 payloads, include/cassie_io_types.h; reference include/<type>.h:20) */
#define CASSIE_USER_IN_T_PACKED_LEN 58
#define CASSIE_OUT_T_PACKED_LEN 697
#define PD_IN_T_PACKED_LEN 476
#define STATE_OUT_T_PACKED_LEN 493

static long long now_usec(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1000000LL + t.tv_nsec / 1000;
}

int main(int argc, char *argv[])
{
    const char *addr = "0.0.0.0", *port = "25000", *model = "../model/cassie.xml", *log_path = NULL, *qlog_path = NULL, *state_path = NULL;
    bool realtime = false, hold = false, pd_mode = false;
    long max_steps = -1;
    int c;
    while ((c = getopt(argc, argv, "a:p:m:rvhl:q:xn:s:")) != -1) {
        switch (c) {
        case 'a': addr = optarg; break;
        case 'p': port = optarg; break;
        case 'm': model = optarg; break;
        case 'r': realtime = true; break;
        case 'v': break;
        case 'h': hold = true; break;
        case 'l': log_path = optarg; break;
        case 'q': qlog_path = optarg; break;
        case 'x': pd_mode = true; break;
        case 'n': max_steps = atol(optarg); break;
        case 's': state_path = optarg; break;
        default:
            printf("Usage: cassiesim [-a address] [-p port] [-m modelfile] [-x] [-r] [-h] [-l log] [-q statelog] [-n steps] [-s statefile]\n"
                   "Simulates the Cassie robot on an MI355X, communicating over UDP (lock step unless -r).\n");
            return 1;
        }
    }
    if (!cassie_mujoco_init(model)) return 2;
    cassie_sim_t *sim = cassie_sim_init(model, false);
    if (!sim) return 2;
    if (hold) cassie_sim_hold(sim);
    if (state_path && access(state_path, R_OK) == 0) {
        cassie_state_t *st = cassie_state_alloc();
        if (cassie_state_load(st, state_path) == 0) { cassie_set_state(sim, st); printf("resumed from %s at t = %.4f\n", state_path, *cassie_sim_time(sim)); }
        else fprintf(stderr, "cassiesim: %s is not a usable state file\n", state_path);
        cassie_state_free(st);
    }

    const int dinlen = pd_mode ? PD_IN_T_PACKED_LEN : CASSIE_USER_IN_T_PACKED_LEN;
    const int doutlen = pd_mode ? STATE_OUT_T_PACKED_LEN : CASSIE_OUT_T_PACKED_LEN;
    const int recvlen = PACKET_HEADER_LEN + dinlen, sendlen = PACKET_HEADER_LEN + doutlen;
    unsigned char *recvbuf = calloc(1, (size_t)recvlen), *sendbuf = calloc(1, (size_t)sendlen);
    const unsigned char *header_in = recvbuf, *data_in = recvbuf + PACKET_HEADER_LEN;
    unsigned char *header_out = sendbuf, *data_out = sendbuf + PACKET_HEADER_LEN;

    cassie_user_in_t user_in;
    cassie_out_t out;
    pd_in_t pd_in;
    state_out_t state_out;
    memset(&user_in, 0, sizeof user_in);
    memset(&pd_in, 0, sizeof pd_in);
    packet_header_info_t hinfo;
    memset(&hinfo, 0, sizeof hinfo);

    int sock = udp_init_host(addr, port);
    if (sock < 0) { fprintf(stderr, "cassiesim: cannot bind %s:%s\n", addr, port); return 3; }
    struct sockaddr_storage src;
    socklen_t srclen = sizeof src;
    FILE *log = log_path ? fopen(log_path, "wb") : NULL, *qlog = qlog_path ? fopen(qlog_path, "wb") : NULL;

    const long long cycle_usec = 1000000 / 2000, timeout_usec = pd_mode ? 100000 : 10000;
    long long send_time = now_usec(), recv_time = now_usec();
    bool run = false;
    long steps = 0, slow = 0;
    printf("Waiting for input...\n");
    fflush(stdout);
    while (max_steps < 0 || steps < max_steps) {
        ssize_t nbytes;
        if (realtime) nbytes = get_newest_packet(sock, recvbuf, (size_t)recvlen, (struct sockaddr *)&src, &srclen);
        else nbytes = wait_for_packet(sock, recvbuf, (size_t)recvlen, (struct sockaddr *)&src, &srclen);
        if (nbytes == recvlen) {
            process_packet_header(&hinfo, header_in, header_out);
            if (pd_mode) unpack_pd_in_t(data_in, &pd_in);
            else unpack_cassie_user_in_t(data_in, &user_in);
            recv_time = now_usec();
            run = true;
        }
        if (!run) continue;
        const long long t0 = now_usec();
        const double sim_t0 = *cassie_sim_time(sim);
        if (pd_mode) { cassie_sim_step_pd(sim, &state_out, &pd_in); pack_state_out_t(&state_out, data_out); }
        else { cassie_sim_step(sim, &out, &user_in); pack_cassie_out_t(&out, data_out); }
        ++steps;
        if (log) { fwrite(data_out, (size_t)doutlen, 1, log); fwrite(data_in, (size_t)dinlen, 1, log); }
        if (qlog) {
            fwrite(cassie_sim_time(sim), sizeof(double), 1, qlog);
            fwrite(cassie_sim_qpos(sim), sizeof(double), 35, qlog);
            fwrite(cassie_sim_qvel(sim), sizeof(double), 32, qlog);
        }
        if (realtime) {
            while (now_usec() - send_time < cycle_usec) {}
            send_time = now_usec();
            if (now_usec() - recv_time > timeout_usec) { memset(&user_in, 0, sizeof user_in); memset(&pd_in, 0, sizeof pd_in); }
        }
        send_packet(sock, sendbuf, (size_t)sendlen, (struct sockaddr *)&src, srclen);
        const double cpu_dt = (double)(now_usec() - t0) / 1e6, sim_dt = *cassie_sim_time(sim) - sim_t0;
        if (cpu_dt > sim_dt + 1e-4) { ++slow; if (realtime) printf("SLOWER THAN REAL TIME BY %6.5fs\n", cpu_dt - sim_dt); }
    }
    printf("%ld steps, %ld of them slower than real time\n", steps, slow);
    if (state_path) {
        cassie_state_t *st = cassie_state_alloc();
        cassie_get_state(sim, st);
        if (cassie_state_save(st, state_path) != 0) fprintf(stderr, "cassiesim: cannot write %s\n", state_path);
        cassie_state_free(st);
    }
    if (log) fclose(log);
    if (qlog) fclose(qlog);
    udp_close(sock);
    cassie_sim_free(sim);
    free(recvbuf); free(sendbuf);
    return 0;
}
