/*
 * udp.c -- non-blocking UDP sockets with newest-packet semantics, as the reference's
 * controller <-> simulator link uses them (reference src/udp.c:58-265).  Kept only so that the
 * drop-in library exports the symbols the ctypes wrapper binds; nothing on the hot path uses it.
 */
#define _GNU_SOURCE
#include "udp.h"

#include <netdb.h>
#include <poll.h>
#include <stdio.h>
#include <string.h>
#include <sys/ioctl.h>
#include <unistd.h>

void process_packet_header(packet_header_info_t *info, const unsigned char *header_in, unsigned char *header_out)
{
    /* byte 0: sender's sequence number, byte 1: the last sequence number the sender saw from us */
    const char seq_in = (char)header_in[0], echoed = (char)header_in[1];
    info->seq_num_out++;
    info->delay = info->seq_num_out - echoed;
    info->seq_num_in_diff = seq_in - info->seq_num_in_last;
    info->seq_num_in_last = seq_in;
    header_out[0] = (unsigned char)info->seq_num_out;
    header_out[1] = (unsigned char)seq_in;
}

static struct addrinfo *resolve(const char *addr, const char *port)
{
    struct addrinfo hints, *res = NULL;
    memset(&hints, 0, sizeof hints);
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = SOCK_DGRAM;
    hints.ai_protocol = IPPROTO_UDP;
    int err = getaddrinfo(addr, port, &hints, &res);
    if (err) {
        printf("%s\n", gai_strerror(err));
        return NULL;
    }
    return res;
}

/* socket bound to `local`, optionally connected to `remote`, switched to non-blocking mode */
static int open_socket(struct addrinfo *local, struct addrinfo *remote)
{
    struct addrinfo *family = remote ? remote : local;
    int sock = socket(family->ai_family, family->ai_socktype, family->ai_protocol);
    if (sock == -1) {
        perror("Error creating socket");
        return -1;
    }
    if (bind(sock, local->ai_addr, local->ai_addrlen)) {
        perror("Error binding to interface address");
        close(sock);
        return -1;
    }
    if (remote && connect(sock, remote->ai_addr, remote->ai_addrlen)) {
        perror("Error connecting to remote address");
        close(sock);
        return -1;
    }
    int nonblocking = 1;
    ioctl(sock, FIONBIO, &nonblocking);
    return sock;
}

int udp_init_host(const char *addr_str, const char *port_str)
{
    struct addrinfo *local = resolve(addr_str, port_str);
    if (!local) return -1;
    int sock = open_socket(local, NULL);
    freeaddrinfo(local);
    return sock;
}

int udp_init_client(const char *remote_addr_str, const char *remote_port_str, const char *local_addr_str,
                    const char *local_port_str)
{
    struct addrinfo *remote = resolve(remote_addr_str, remote_port_str);
    if (!remote) return -1;
    struct addrinfo *local = resolve(local_addr_str, local_port_str);
    if (!local) {
        freeaddrinfo(remote);
        return -1;
    }
    int sock = open_socket(local, remote);
    freeaddrinfo(remote);
    freeaddrinfo(local);
    return sock;
}

void udp_close(int sock) { close(sock); }

ssize_t get_newest_packet(int sock, void *recvbuf, size_t recvlen, struct sockaddr *src_addr, socklen_t *addrlen)
{
    /* drain the receive queue: keep the last datagram of exactly the expected size, drop the others */
    ssize_t got = -1;
    struct pollfd fd;
    fd.fd = sock;
    fd.events = POLLIN;
    fd.revents = 0;
    while (poll(&fd, 1, 0)) {
        int avail = 0;
        ioctl(sock, FIONREAD, &avail);
        if ((size_t)avail == recvlen) got = recvfrom(sock, recvbuf, recvlen, 0, src_addr, addrlen);
        else recv(sock, recvbuf, 0, 0);
    }
    return got;
}

ssize_t wait_for_packet(int sock, void *recvbuf, size_t recvlen, struct sockaddr *src_addr, socklen_t *addrlen)
{
    ssize_t got;
    do {
        struct pollfd fd;
        fd.fd = sock;
        fd.events = POLLIN;
        fd.revents = 0;
        while (!poll(&fd, 1, 0)) {
        }
        got = get_newest_packet(sock, recvbuf, recvlen, src_addr, addrlen);
    } while (got != (ssize_t)recvlen);
    return got;
}

ssize_t send_packet(int sock, void *sendbuf, size_t sendlen, struct sockaddr *dst_addr, socklen_t addrlen)
{
    ssize_t sent;
    do {
        sent = sendto(sock, sendbuf, sendlen, 0, dst_addr, addrlen);
    } while (sent == -1);
    return sent;
}
